"""Golden vectors for beam search: transformers' OWN `GenerationMixin.generate` driven the way the reference drives it
(reference modeling/wrapper.py:306-313,443-451: num_beams = num_return_sequences = k, barebones GenerationConfig with
bos / decoder_start / eos / forced_eos / pad ids and max_length, use_cache=False, library defaults otherwise:
length_penalty 1.0, early_stopping False, do_sample False), on a STUB encoder-decoder whose logits are a seeded table
lookup of (sample, position, previous token, prefix hash).  The same tables drive `multimodalanalytical_amd.beam`
in tests/test_beam_cpu.py; sequences and sequence scores must agree.

Run in the build container:  python oracle/make_beam_goldens.py   ->  tests/golden/beam_cases.npz
Version skew: the reference pins transformers 4.48.3 (uv.lock); this container has the version printed into the
fixture's meta (5.x: beam search re-implemented in vectorised form, same algorithm and outputs by its own tests).
The reference's model cannot be run through generate() here, hence the stub.
"""
import json
import os

import numpy as np
import torch
import transformers
from transformers import GenerationConfig, GenerationMixin, PretrainedConfig, PreTrainedModel
from transformers.modeling_outputs import BaseModelOutput, Seq2SeqLMOutput

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
PAD, BOS, EOS = 0, 2, 3
NHASH = 61


def table_logits(t1, t2, sid, prefix):
    """Logits of the NEXT token for every row: prefix (R, t) int64, sid (R,) -> (R, V).
    t1[sid, t-1, last token] + t2[sid, polynomial hash of the prefix]."""
    t = prefix.shape[1]
    w = torch.arange(1, t + 1, dtype=torch.long) * 7 + 3
    h = (prefix * w).sum(1) % NHASH
    return t1[sid, t - 1, prefix[:, -1]] + t2[sid, h]


class StubConfig(PretrainedConfig):
    model_type = "afm_beam_stub"

    def __init__(self, vocab_size=12, **kw):
        super().__init__(is_encoder_decoder=True, vocab_size=vocab_size, **kw)


class _Enc(torch.nn.Module):
    main_input_name = "input_ids"

    def forward(self, *a, **k):
        raise RuntimeError("encoder outputs are passed in")


class Stub(PreTrainedModel, GenerationMixin):
    config_class = StubConfig
    main_input_name = "input_ids"

    def __init__(self, config, t1, t2):
        super().__init__(config)
        self.t1, self.t2 = t1, t2
        self.dummy = torch.nn.Parameter(torch.zeros(1))
        self.encoder = _Enc()

    def get_encoder(self):
        return self.encoder

    def forward(self, input_ids=None, attention_mask=None, decoder_input_ids=None, encoder_outputs=None, **kw):
        hs = encoder_outputs[0] if not isinstance(encoder_outputs, dict) else encoder_outputs["last_hidden_state"]
        sid = hs[:, 0, 0].long()                              # the sample id travels in the (beam-expanded) encoder state
        R, T = decoder_input_ids.shape
        logits = torch.stack([table_logits(self.t1, self.t2, sid, decoder_input_ids[:, :t + 1]) for t in range(T)], 1)
        return Seq2SeqLMOutput(logits=logits)

    def prepare_inputs_for_generation(self, decoder_input_ids, encoder_outputs=None, attention_mask=None, **kw):
        return {"decoder_input_ids": decoder_input_ids, "encoder_outputs": encoder_outputs, "attention_mask": attention_mask}


CASES = [  # name, B, k, V, max_length, eos bias, scale
    ("b3k3", 3, 3, 12, 12, 0.0, 1.5), ("b4k5", 4, 5, 16, 16, 0.5, 1.0), ("b2k10", 2, 10, 12, 10, -0.5, 2.0),
    ("b5k2_long", 5, 2, 10, 24, -1.0, 1.0), ("b2k4_early", 2, 4, 12, 14, 2.0, 1.0), ("b1k30", 1, 30, 40, 12, 0.0, 1.0),
]


def main():
    out = {}
    meta = {"transformers": transformers.__version__, "reference_pin": "4.48.3", "pad": PAD, "bos": BOS, "eos": EOS,
            "nhash": NHASH, "cases": {}}
    for i, (name, B, k, V, L, eos_bias, scale) in enumerate(CASES):
        g = torch.Generator().manual_seed(100 + i)
        t1 = torch.randn(B, L, V, V, generator=g) * scale
        t2 = torch.randn(B, NHASH, V, generator=g) * scale
        t1[..., EOS] += eos_bias
        t1[..., PAD] = -1e4; t1[..., BOS] = -1e4; t2[..., PAD] = 0; t2[..., BOS] = 0    # pad / bos are never generated
        model = Stub(StubConfig(vocab_size=V), t1, t2).eval()
        gen_cfg = GenerationConfig(bos_token_id=BOS, decoder_start_token_id=BOS, eos_token_id=EOS, forced_eos_token_id=EOS,
                                   max_length=L, pad_token_id=PAD)
        hs = torch.arange(B, dtype=torch.float32).view(B, 1, 1).repeat(1, 2, 1)
        enc = BaseModelOutput(last_hidden_state=hs)
        with torch.no_grad():
            res = model.generate(encoder_outputs=enc, attention_mask=torch.ones(B, 2, dtype=torch.long), num_beams=k,
                                 num_return_sequences=k, generation_config=gen_cfg, use_cache=False,
                                 return_dict_in_generate=True, output_scores=True)
        seqs = res.sequences
        out[f"{name}/t1"], out[f"{name}/t2"] = t1.numpy(), t2.numpy()
        out[f"{name}/sequences"] = seqs.numpy()
        out[f"{name}/sequences_scores"] = res.sequences_scores.numpy()
        meta["cases"][name] = {"B": B, "k": k, "V": V, "max_length": L}
        print(name, tuple(seqs.shape), res.sequences_scores[:k].tolist()[:3])
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "beam_cases.npz"), **out)


if __name__ == "__main__":
    main()
