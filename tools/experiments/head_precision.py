"""how much of the fp16 mode's logits error is made in the LAST stages (final decoder LayerNorm output rounded to fp16,
token_ff in fp16)?  Fresh-init c2 / c4 at B = 2 against the CPU oracle: the engine's logits, and the logits recomputed from the engine's own
fp32 stream in front of the final LayerNorm with an fp32 LayerNorm + fp32 token_ff (torch, experiment only).
    python tools/experiments/head_precision.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_gpu_shapes import _case, _to

DEV = "cuda:0"


def main():
    from multimodalanalytical_amd.engine import Seq2SeqEngine
    for name in ("c2", "c3", "c4"):
        wl, cfg, inputs, sd, ref, _ = _case(name)
        eng = Seq2SeqEngine(cfg, wl["data"], "Smiles", wl["data"]["Smiles"]["vocab_size"], device=DEV, compute_dtype=torch.float16, seed=5)
        eng.load_state_dict(sd)
        eng.eval()
        cap = {}
        orig = eng._ln_fwd

        def spy(x, prefix, saved, key, out_dtype=None, pend=None):
            y, xs = orig(x, prefix, saved, key, out_dtype=out_dtype, pend=pend)
            if prefix == "decoder.norm.":
                cap["xs"] = xs.clone()
            if prefix == "encoder.norm.":
                cap["mem_in"] = xs.clone()
            return y, xs
        eng._ln_fwd = spy
        enc, am, dec, dm, labels = inputs
        out = eng.forward(_to(enc), am.to(DEV), dec.to(DEV), dm.to(DEV), None)
        rl = ref["logits"].double()
        scale = float(rl.abs().max())
        err = float((out["logits"].cpu().double() - rl).abs().max()) / scale
        xs = cap["xs"].double()
        g, b = sd["decoder.norm.weight"].to(DEV).double(), sd["decoder.norm.bias"].to(DEV).double()
        mu, var = xs.mean(1, keepdim=True), xs.var(1, unbiased=False, keepdim=True)
        hf = (xs - mu) / torch.sqrt(var + 1e-5) * g + b
        W, bb = sd["token_ff.weight"].to(DEV).double(), sd["token_ff.bias"].to(DEV).double()
        lg = (hf @ W.T + bb).view(rl.shape).cpu()
        err_head = float((lg - rl).abs().max()) / scale
        # the same with hf rounded to fp16 but W / product in fp64 (what a split-pair W alone would buy), and with W rounded only
        lg_h = ((hf.half().double()) @ W.T + bb).view(rl.shape).cpu()
        lg_w = (hf @ W.half().double().T + bb).view(rl.shape).cpu()
        print(f"{name}: engine {err:.3e}   exact head on the engine's stream {err_head:.3e}   hf fp16 only {float((lg_h - rl).abs().max()) / scale:.3e}   "
              f"W fp16 only {float((lg_w - rl).abs().max()) / scale:.3e}", flush=True)


if __name__ == "__main__":
    main()
