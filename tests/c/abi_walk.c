/* C caller of libishara_hip.so: proves the boundary is a plain C ABI (include/ishara_hip.h compiles as C, no torch / C++ types in the
 * signatures) and walks the host-only entry points — create each of the three families, list the flat parameter layout, audit the
 * workspace plan, destroy.  No GPU is touched.  Built and run by tests/test_abi.py with gcc + dlopen. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include "ishara_hip.h"

typedef int (*create_fn)(const ishara_config*, ishara_model**);
typedef void (*destroy_fn)(ishara_model*);
typedef int64_t (*i64_fn)(const ishara_model*);
typedef int32_t (*i32_fn)(const ishara_model*);
typedef int (*info_fn)(const ishara_model*, int32_t, const char**, int32_t*, int64_t*, int64_t*, int32_t*);
typedef const char* (*err_fn)(void);

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: abi_walk <libishara_hip.so>\n"); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    create_fn create = (create_fn)dlsym(h, "ishara_create");
    destroy_fn destroy = (destroy_fn)dlsym(h, "ishara_destroy");
    i64_fn total = (i64_fn)dlsym(h, "ishara_param_total"), train = (i64_fn)dlsym(h, "ishara_param_trainable"), wsb = (i64_fn)dlsym(h, "ishara_workspace_bytes");
    i32_fn entries = (i32_fn)dlsym(h, "ishara_param_entries"), plan = (i32_fn)dlsym(h, "ishara_workspace_plan_check"), frames = (i32_fn)dlsym(h, "ishara_encoder_output_frames");
    info_fn info = (info_fn)dlsym(h, "ishara_param_info");
    err_fn last_error = (err_fn)dlsym(h, "ishara_last_error");
    if (!create || !destroy || !total || !train || !wsb || !entries || !plan || !frames || !info || !last_error) { fprintf(stderr, "missing symbol\n"); return 2; }
    for (int family = 0; family < 3; ++family) {
        ishara_config c;
        memset(&c, 0, sizeof c);
        c.family = family;
        c.dim = 256; c.num_heads = 8; c.expansion_factor = family == 2 ? 4 : 2; c.transformer_kernel_size = 15; c.dropout_rate = 0.1f;
        c.num_conv_squeeze_blocks = 2; c.num_conv_conform_blocks = family == 2 ? 4 : 2; c.num_conv_per_block = 3;
        c.num_kernel_sizes = 3; c.kernel_sizes[0] = 11; c.kernel_sizes[1] = 5; c.kernel_sizes[2] = 3;
        c.frames = 384; c.features = family == 2 ? 80 : 224; c.num_classes = 60; c.head_dropout = 0.4f; c.conformer_attn_dropout = 0.1f;
        c.dtype = ISHARA_BF16; c.max_batch = 4; c.max_label_len = 64; c.attn_impl = 1;
        c.reduce_layer_index = 1; c.recover_layer_index = 3; c.half_step_residual = 1;
        ishara_model* m = 0;
        if (create(&c, &m) != 0) { fprintf(stderr, "family %d: %s\n", family, last_error()); return 1; }
        const int32_t n = entries(m);
        int64_t sum = 0, prev_end = -1;
        for (int32_t i = 0; i < n; ++i) {
            const char* name; int32_t nd, tr; int64_t shape[2], off;
            if (info(m, i, &name, &nd, shape, &off, &tr) != 0) return 1;
            sum += shape[0] * (nd == 2 ? shape[1] : 1);
            (void)prev_end;
        }
        if (info(m, n, 0, 0, 0, 0, 0) == 0) { fprintf(stderr, "out-of-range index accepted\n"); return 1; }
        if (sum != total(m) || train(m) > total(m) || plan(m) <= 0 || wsb(m) <= 0) { fprintf(stderr, "family %d: inconsistent layout\n", family); return 1; }
        printf("family %d: %d entries, %lld parameters (%lld trainable), workspace %lld bytes, encoder frames %d\n", family, (int)n, (long long)total(m),
               (long long)train(m), (long long)wsb(m), family ? (int)frames(m) : 0);
        destroy(m);
    }
    dlclose(h);
    return 0;
}
