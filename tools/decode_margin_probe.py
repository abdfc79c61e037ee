"""How many train steps on one repeated batch until every frame's top-2 margin clears the 16-bit logit error?  (Tunes
tests/test_decode_confident_gpu.py.)  Prints per checkpoint: loss, per-clip minimum margin of the eval-mode logits, decode lengths."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ishara_amd import get_model
from oracle import ishara_oracle as O

KW = dict(dim=64, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1, input_shape=(176, 276))
B = 8
ocfg = O.Config(**KW, dropout_rate=0.0, head_dropout=0.0, conformer_attn_dropout=0.0)
for lr in (4e-3, 2e-3):
    model = get_model(**KW, dropout_rate=0.0, head_dropout=0.0, conformer_attn_dropout=0.0, dtype="bf16", max_batch=B, seed=21)
    x, y = O.synthetic_batch(ocfg, B, seed=9)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    model.optimizer.learning_rate = lr
    for step in range(1, 1501):
        loss = model.train_on_batch(xd, yd)
        if step in (1, 100, 200, 300, 400, 600, 800, 1000, 1500):
            lg = model(xd, training=False).cpu().numpy()
            top2 = np.sort(lg, -1)[..., -2:]
            mg = (top2[..., 1] - top2[..., 0])
            dec = model.decode_batch(torch.from_numpy(lg).cuda())
            lab = [int((y[b] != 59).sum()) for b in range(B)]
            ok = [bool(np.array_equal(dec[b], y[b][:lab[b]])) for b in range(B)]
            print(f"lr {lr} step {step}: loss {float(loss.item()):.4f} min-margin/clip {np.round(mg.min(1), 3).tolist()} frac>0.3 {float((mg > 0.3).mean()):.3f} "
                  f"declen {[len(d) for d in dec]} labels {lab} exact {ok}", flush=True)
