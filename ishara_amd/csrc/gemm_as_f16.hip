// The A-stationary NT GEMM (gemm_as.hip) instantiated for fp16 operands — the ISHARA_F16 inference path (BASELINE configs[4]).
// Same kernel source: the operand element type, the 8-element vector type and the MFMA builtin are macros of that file; only the
// forward feature combinations are compiled (fp16 is refused for training).
#define AS_F16 1
#include "gemm_as.hip"
