#!/usr/bin/env python3
"""BASELINE configs[4]: single-clip greedy-decode latency (B=1, T=384) of the TFLite-shaped wrapper:
preprocess + encoder forward + decode, eager launches vs one hipGraph replay.  Prints one JSON line."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from ishara_amd import get_model
from ishara_amd.tflite_model import TFLiteModel

model = get_model(dim=256, num_conv_squeeze_blocks=2, num_conv_conform_blocks=2, kernel_sizes=[11, 5, 3], input_shape=(384, 276),
                  dtype="bf16", max_batch=1, seed=0)
x = np.random.default_rng(0).standard_normal((300, 276)).astype(np.float32)
res = {}
for name, g in (("eager", False), ("hipgraph", True)):
    t = TFLiteModel(model, max_frames=1024, use_graph=g)
    for _ in range(5): t(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): t(x)
    torch.cuda.synchronize(); res[name + "_ms_per_clip_incl_h2d_d2h"] = (time.perf_counter() - t0) / 50 * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): (t._graph.replay() if g else t._launch())
    e1.record(); torch.cuda.synchronize(); res[name + "_ms_device_only"] = e0.elapsed_time(e1) / 50
print(json.dumps(dict(metric="single-clip latency, B=1 T=384 d256 2+2 bf16 (preprocess+forward+greedy decode)", unit="ms", **res,
                      reference="TFLite CPU fp16-weights: 107-262 ms per clip for sibling models (BASELINE.md)")))
