"""Forward + backward of the c2 engine with dropout: keep-bit path vs re-hash path, run twice each; reports where logits / loss differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import synth
from multimodalanalytical_amd.engine import Seq2SeqEngine
from multimodalanalytical_amd.params import ParamStore, build_specs
from multimodalanalytical_amd.x2 import X2
from oracle import afm_oracle as O

DEV = "cuda:0"
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
cd = {"bf16": torch.bfloat16, "bf16x3": X2.dtype}[mode]
wl = synth.WORKLOADS["c2"]
batch, _ = synth.make_batch("c2", 2, seed=11)
inputs = O.batch_to_model_inputs(batch, "Smiles")
cfg = dict(wl["cfg"], dropout=0.1)
V = wl["data"]["Smiles"]["vocab_size"]
ps = ParamStore(build_specs(cfg, wl["data"], V), "cpu", False)
ps.init_(5)
sd = {k: v.clone() for k, v in ps.state_dict().items()}
if cfg["positional_encoding_type"] == "sin_cos":
    sd["embedding.positional_encodings.pos_enc"] = O.sincos_table(cfg["d_model"], cfg["max_position_embeddings"])


def to(x):
    return {k: to(v) for k, v in x.items()} if isinstance(x, dict) else x.to(DEV)


res = []
for kb in (True, True, False, False):
    eng = Seq2SeqEngine(cfg, wl["data"], "Smiles", V, device=DEV, compute_dtype=cd, seed=5)
    eng.keep_bits = kb
    eng.load_state_dict(sd)
    eng.train()
    enc, am, dec, dm, labels = inputs
    out = eng.forward(to(enc), am.to(DEV), dec.to(DEV), dm.to(DEV), labels.to(DEV), backward=True)
    torch.cuda.synchronize()
    res.append((kb, out["logits"].float().cpu().clone(), float(out["loss"]), {k: eng.ps.g(k).cpu().clone() for k in sd if not k.endswith("pos_enc")}))
for i in range(1, 4):
    a, b = res[0], res[i]
    print(f"run0(bits={a[0]}) vs run{i}(bits={b[0]}): logits max diff {float((a[1] - b[1]).abs().max()):.3e}, loss {a[2]!r} vs {b[2]!r}",
          "worst grad rel diff", max(((float((a[3][k] - b[3][k]).norm()) / (float(b[3][k].norm()) + 1e-30)), k) for k in a[3]))
print("run2 vs run3 (both re-hash): logits", float((res[2][1] - res[3][1]).abs().max()), res[2][2], res[3][2],
      "worst grad rel diff", max(((float((res[2][3][k] - res[3][3][k]).norm()) / (float(res[3][3][k].norm()) + 1e-30)), k) for k in res[2][3]))
