// Shared device/host helpers for the ishara_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef _Float16 f16;                                      // inference-only storage type (BASELINE configs[4]: fp16 B=1 decode)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define WAVE 64
#define DEVI __device__ __forceinline__

template <typename T> struct is_bf16_t { static constexpr bool value = false; };
template <> struct is_bf16_t<bf16> { static constexpr bool value = true; };

DEVI float to_f(float v) { return v; }
DEVI float to_f(bf16 v) { return (float)v; }
DEVI float to_f(f16 v) { return (float)v; }
// 16-bit storage types (8 elements per 16-byte chunk)
template <typename T> struct is_16b_t { static constexpr bool value = false; };
template <> struct is_16b_t<bf16> { static constexpr bool value = true; };
template <> struct is_16b_t<f16> { static constexpr bool value = true; };
template <typename T> DEVI T from_f(float v);
template <> DEVI float from_f<float>(float v) { return v; }
template <> DEVI bf16 from_f<bf16>(float v) { return (bf16)v; }   // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
template <> DEVI f16 from_f<f16>(float v) { return (f16)v; }

// ---- 8-element vector IO (16 B for bf16, 2x16 B for f32) -------------------
DEVI void load8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
DEVI void load8(const bf16* p, float (&v)[8]) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
DEVI void load8(const f16* p, float (&v)[8]) {
    const f16x8 a = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
DEVI void store8(f16* p, const float (&v)[8]) {
    f16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (f16)v[i];
    *reinterpret_cast<f16x8*>(p) = a;
}
DEVI void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
DEVI void store8(bf16* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16)v[i];
    *reinterpret_cast<bf16x8*>(p) = a;
}
// bounds-aware: n = number of valid elements (<=8); requires p 16-B aligned when n==8
template <typename T> DEVI void load8_n(const T* p, float (&v)[8], int n) {
    if (n >= 8) { load8(p, v); return; }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (i < n) ? to_f(p[i]) : 0.f;
}
template <typename T> DEVI void store8_n(T* p, const float (&v)[8], int n) {
    if (n >= 8) { store8(p, v); return; }
#pragma unroll
    for (int i = 0; i < 8; ++i) if (i < n) p[i] = from_f<T>(v[i]);
}
DEVI void load4(const float* p, float (&v)[4]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}

DEVI void load4g(const float* p, float (&v)[4]) { load4(p, v); }
DEVI void load4g(const bf16* p, float (&v)[4]) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xFFFF0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xFFFF0000u);
}
DEVI void load4g(const f16* p, float (&v)[4]) {
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
    const f16x4 a = *reinterpret_cast<const f16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)a[i];
}
DEVI void store4(f16* p, const float (&v)[4]) {
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
    f16x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = (f16)v[i];
    *reinterpret_cast<f16x4*>(p) = a;
}
DEVI void store4(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
DEVI void store4(bf16* p, const float (&v)[4]) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = (bf16)v[i];
    *reinterpret_cast<bf16x4*>(p) = a;
}

// ---- math ---------------------------------------------------------------------
// v_exp_f32 + v_rcp_f32 (1 ulp each): an IEEE division costs ~10 more VALU instructions per element in every Swish / GLU / gate
DEVI float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
DEVI float swishf_(float x) { return x * sigmoidf_(x); }
DEVI float dswishf_(float x) { const float s = sigmoidf_(x); return s * (1.f + x * (1.f - s)); }

// ---- wave / block reductions ----------------------------------------------------
DEVI float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
DEVI float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- counter-based dropout RNG (mirrored by oracle/rng.py) ----------------------
__host__ __device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t rng_site_key(uint32_t seed, uint32_t site) {
    return lowbias32(seed ^ (site * 0x9E3779B9u));
}
__host__ __device__ __forceinline__ uint32_t rng_row_key(uint32_t site_key, uint32_t row) {
    return lowbias32(site_key ^ (row * 0x85EBCA6Bu));
}
// One hash serves a PAIR of columns (2k, 2k+1): 16 bits each, low half for the even column.  The hash (two quarter-rate
// 32-bit multiplies) is 60 % of the VALU work of an attention score; kernels whose lanes own consecutive columns call
// rng_pair once per pair.  keep iff the column's 16 bits >= thr16 = round(rate * 65536): P(keep) = 1 - thr16 / 65536.
__host__ __device__ __forceinline__ uint32_t rng_pair(uint32_t row_key, uint32_t col) {
    return lowbias32(row_key ^ (col >> 1));
}
__host__ __device__ __forceinline__ bool rng_keep(uint32_t row_key, uint32_t col, uint32_t thr) {
    const uint32_t h = rng_pair(row_key, col);
    return ((col & 1u) ? (h >> 16) : (h & 0xffffu)) >= thr;
}
// Attention-PROBABILITY dropout (the [T,T] score matrices: 302 M decisions per layer in config #2, where the hash was 40 % of the forward
// kernel's VALU issue slots) draws FOUR decisions from one hash: keys 4k .. 4k+3 of a query row share lowbias32(row_key ^ k), key 4k+e
// takes byte e.  keep iff byte >= thr8 = round(rate * 256): P(keep) = 1 - thr8/256 (rate 0.2 -> 0.80078, 0.1 -> 0.89844), and the kept
// probabilities are scaled by 256/(256 - thr8) = 1/P(keep), so the expectation is exact for the rate actually drawn.  Mirrored by
// oracle/rng.py (keep_mask_attn).
__host__ __device__ __forceinline__ uint32_t rng_quad(uint32_t row_key, uint32_t col) {
    return lowbias32(row_key ^ (col >> 2));
}
__host__ __device__ __forceinline__ bool rng_keep_q(uint32_t row_key, uint32_t col, uint32_t thr8) {
    return ((rng_quad(row_key, col) >> (8u * (col & 3u))) & 0xffu) >= thr8;
}
#if defined(__HIPCC__)
// keep flags (bit e) of the 4 consecutive keys starting at a multiple of 4: ONE hash
__device__ __forceinline__ uint32_t rng_bits4_q(uint32_t row_key, uint32_t col4, uint32_t thr8) {
    const uint32_t h = rng_quad(row_key, col4);
    return ((h & 0xffu) >= thr8 ? 1u : 0u) | (((h >> 8) & 0xffu) >= thr8 ? 2u : 0u) | (((h >> 16) & 0xffu) >= thr8 ? 4u : 0u) | ((h >> 24) >= thr8 ? 8u : 0u);
}
// keep flags (bit e) of the 4 consecutive columns starting at a multiple of 4: two hashes
__device__ __forceinline__ uint32_t rng_bits4(uint32_t row_key, uint32_t col4, uint32_t thr) {
    const uint32_t h0 = rng_pair(row_key, col4), h1 = rng_pair(row_key, col4 + 2);
    return ((h0 & 0xffffu) >= thr ? 1u : 0u) | ((h0 >> 16) >= thr ? 2u : 0u) | ((h1 & 0xffffu) >= thr ? 4u : 0u) | ((h1 >> 16) >= thr ? 8u : 0u);
}
// inverted dropout of N consecutive columns starting at an EVEN column: one hash per pair
template <int N>
__device__ __forceinline__ void rng_apply(uint32_t row_key, uint32_t col_even, uint32_t thr, float scale, float (&v)[N]) {
    static_assert(N % 2 == 0, "pairs");
#pragma unroll
    for (int e = 0; e < N; e += 2) {
        const uint32_t h = rng_pair(row_key, col_even + e);
        v[e] = (h & 0xffffu) >= thr ? v[e] * scale : 0.f;
        v[e + 1] = (h >> 16) >= thr ? v[e + 1] * scale : 0.f;
    }
}
#endif
static inline uint32_t rng_threshold(float rate) {
    double t = (double)rate * 65536.0 + 0.5;
    if (t < 0) t = 0;
    if (t > 65535.0) t = 65535.0;
    return (uint32_t)t;
}
struct DropSpec {          // one dropout application
    uint32_t key;          // rng_site_key(seed, site)
    uint32_t thr;          // 16-bit threshold: keep iff the column's half of the pair hash >= thr ; thr==0 -> disabled
    float scale;           // 1/(1-rate)
};
static inline DropSpec make_drop(uint32_t seed, uint32_t site, float rate, bool training) {
    DropSpec d;
    d.key = rng_site_key(seed, site);
    d.thr = (training && rate > 0.f) ? rng_threshold(rate) : 0u;
    d.scale = (training && rate > 0.f) ? 1.0f / (1.0f - rate) : 1.0f;
    return d;
}

// attention-probability sites: 8-bit threshold, scale = 1 / P(keep) of the quantised rate (see rng_quad)
static inline DropSpec make_drop_attn(uint32_t seed, uint32_t site, float rate, bool training) {
    DropSpec d;
    d.key = rng_site_key(seed, site);
    double t = (double)rate * 256.0 + 0.5;
    uint32_t thr8 = (training && rate > 0.f) ? (uint32_t)(t < 0 ? 0 : (t > 255.0 ? 255.0 : t)) : 0u;
    d.thr = thr8;
    d.scale = thr8 ? 256.0f / (256.0f - (float)thr8) : 1.0f;
    return d;
}

#define HIP_CHECK_RET(expr)                                                         \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) { ishara_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); return -2; } \
    } while (0)

void ishara_set_error(const char* fmt, ...);
