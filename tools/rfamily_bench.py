#!/usr/bin/env python3
"""Training pass (forward + backward through ishara_encoder_forward / _backward) of the two torch encoder families at the reference's own default
shapes, for a per-kernel table under rocprofv3 (`rocprofv3 --kernel-trace --stats -- python tools/rfamily_bench.py squeezeformer`):
  squeezeformer: SqueezeformerEncoder(input_dim 80, encoder_dim 512, 16 layers, reduce 7 / recover 15, 8 heads, ffn x4, conv k 31) —
                 squeezeformer/encoder.py:30-46 — on B clips of T = 800 frames (199 frames after the conv2d subsampling)
  conformer:     ConformerEncoder(dim 512, 6 layers, 8 heads, ffn x4, k 31) — conformer/conformer.py:89-90 — on T = 384
Prints ms per forward and per forward+backward (HIP events on the current stream); synthetic inputs, dropout at the reference default 0.1."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
kind = sys.argv[1] if len(sys.argv) > 1 else "squeezeformer"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
if kind == "squeezeformer":
    from ishara_amd import SqueezeformerEncoder
    T, F = 800, 80
    enc = SqueezeformerEncoder(F, 512, 16, 7, 15, 8, 4, 2, 0.1, 0.1, 0.1, 0.1, 31, False, seq_len=T, max_batch=B, dtype="bf16")
else:
    from ishara_amd import ConformerEncoder
    T, F = 384, 512
    enc = ConformerEncoder(512, 6, 8, 4, 31, 0.1, seq_len=T, max_batch=B, dtype="bf16")
x = torch.randn(B, T, F, device="cuda")
enc.train()
def ev(): return torch.cuda.Event(enable_timing=True)
def fwd_bwd():
    xi = x.clone().requires_grad_(True)
    y = enc(xi)
    y.backward(torch.ones_like(y))
for _ in range(2): fwd_bwd()
torch.cuda.synchronize()
e0, e1, e2 = ev(), ev(), ev()
e0.record()
with torch.no_grad():
    for _ in range(steps): enc(x)
e1.record()
for _ in range(steps): fwd_bwd()
e2.record(); torch.cuda.synchronize()
print(json.dumps(dict(family=kind, batch=B, frames_in=T, frames_out=enc.T_out, features=F, params=enc.n_train, dtype="bf16",
                      ms_forward=e0.elapsed_time(e1) / steps, ms_forward_backward=e1.elapsed_time(e2) / steps,
                      input_frames_per_s_training=B * T / (e1.elapsed_time(e2) / steps * 1e-3))))
