"""Host-side beam bookkeeping for `HFWrapper.generate(n_beams > 1)`.

Restates the control flow HF `GenerationMixin._beam_search` + `BeamSearchScorer` follow when called as
the reference calls them (reference modeling/wrapper.py:443-451: num_beams = num_return_sequences =
n_beams, length_penalty 1.0, early_stopping False, forced EOS at max_length, pad id 0): per step
log-softmax + running beam score, top 2k candidates per sample, EOS candidates inside the top k close
a hypothesis scored sum_logprobs / generated_len, the first k non-EOS candidates continue; a sample
is done when it holds k hypotheses and its worst kept score beats best_running / generated_len.
The arithmetic (decoder, logits) runs on the GPU; this module only moves 2k (score, token, beam)
triples per sample per step to the host, as HF's Python loop does.

Parity status: PINNED to transformers' own `generate` on a table-lookup stub model (oracle/make_beam_goldens.py ->
tests/golden/beam_cases.npz, tests/test_beam_cpu.py): sequences and sequence scores equal HF's on every case.
Version skew, stated: the build container has transformers 5.x, the reference pins 4.48.3.  The two differ in ONE
rule, the per-sample stop test with early_stopping=False: 4.48.3's BeamSearchScorer.process compares the worst kept
hypothesis with the best of ALL 2k candidates of the step (`next_scores[batch_idx].max()`, EOS candidates included),
5.x's vectorised `_check_early_stop_heuristic` with the best RUNNING beam (`running_beam_scores[:, :1]`); the 5.x rule
can stop a sample one or more steps earlier when EOS candidates dominate.  `stop_rule="hf5"` is the rule the fixture
pins (all cases equal); `stop_rule="hf4"`, the default, is the reference's pinned version (equal on every case where the
two rules coincide, never worse in score where they do not).
"""
from __future__ import annotations

from typing import List, Tuple

import torch


class BeamHypotheses:
    def __init__(self, num_beams: int, length_penalty: float = 1.0):
        self.num_beams, self.length_penalty = num_beams, length_penalty
        self.beams: List[Tuple[float, List[int]]] = []
        self.worst_score = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp: List[int], sum_logprobs: float, generated_len: int) -> None:
        score = sum_logprobs / (generated_len ** self.length_penalty)
        if len(self) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self) > self.num_beams:
                srt = sorted((s, i) for i, (s, _) in enumerate(self.beams))
                del self.beams[srt[0][1]]
                self.worst_score = srt[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs: float, generated_len: int) -> bool:
        if len(self) < self.num_beams:
            return False
        return self.worst_score >= best_sum_logprobs / (generated_len ** self.length_penalty)


def beam_search(step_fn, reorder_fn, B: int, k: int, V: int, max_length: int, bos: int, eos: int, pad: int,
                device, stop_rule: str = "hf4") -> Tuple[torch.Tensor, torch.Tensor]:
    """step_fn(ids_last (B*k,)) -> fp32 logits (B*k, V) for the next position; reorder_fn(beam_idx (B*k,)).
    Returns (sequences (B*k, L) int64 best-first per sample, scores (B*k,))."""
    ids = torch.full((B * k, 1), bos, dtype=torch.long, device=device)
    beam_scores = torch.zeros(B, k, dtype=torch.float32, device=device)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    hyps = [BeamHypotheses(k) for _ in range(B)]
    done = [False] * B
    while True:
        cur_len = ids.shape[1]
        logits = step_fn(ids[:, -1])
        logp = torch.log_softmax(logits.float(), dim=-1)
        if cur_len == max_length - 1:                       # ForcedEOSTokenLogitsProcessor
            forced = torch.full_like(logp, float("-inf"))
            forced[:, eos] = 0.0
            logp = forced
        scores = (logp + beam_scores[:, None]).view(B, k * V)
        top_s, top_i = torch.topk(scores, 2 * k, dim=1, largest=True, sorted=True)
        top_s_h, top_i_h = top_s.cpu(), top_i.cpu()
        ids_h = ids.cpu()
        nxt_scores = torch.zeros(B, k)
        nxt_tokens = torch.zeros(B, k, dtype=torch.long)
        nxt_index = torch.zeros(B, k, dtype=torch.long)
        for b in range(B):
            if done[b]:
                nxt_tokens[b] = pad
                nxt_index[b] = b * k
                continue
            j = 0
            for rank in range(2 * k):
                s, idx = float(top_s_h[b, rank]), int(top_i_h[b, rank])
                beam, tok = idx // V, idx % V
                row = b * k + beam
                if tok == eos:
                    if rank >= k:
                        continue
                    hyps[b].add(ids_h[row].tolist(), s, generated_len=cur_len)   # cur_len+1 tokens, minus the prompt
                else:
                    nxt_scores[b, j], nxt_tokens[b, j], nxt_index[b, j] = s, tok, row
                    j += 1
                if j == k:
                    break
            best = float(top_s_h[b].max()) if stop_rule == "hf4" else float(nxt_scores[b, 0])
            done[b] = done[b] or hyps[b].is_done(best, cur_len)
        beam_scores = nxt_scores.view(-1).to(device)
        beam_idx = nxt_index.view(-1).to(device)
        ids = torch.cat([ids.index_select(0, beam_idx), nxt_tokens.view(-1, 1).to(device)], dim=1)
        reorder_fn(beam_idx)
        if all(done) or ids.shape[1] >= max_length:
            break
    # finalize: open beams of unfinished samples become hypotheses
    ids_h, bs_h = ids.cpu(), beam_scores.cpu()
    for b in range(B):
        if done[b]:
            continue
        for j in range(k):
            row = b * k + j
            hyps[b].add(ids_h[row].tolist(), float(bs_h[row]), generated_len=ids_h.shape[1] - 1)
    best, best_scores = [], []
    for b in range(B):
        srt = sorted(hyps[b].beams, key=lambda x: x[0])
        for _ in range(k):
            s, h = srt.pop()
            best.append(h); best_scores.append(s)
    L = min(max(len(h) for h in best) + 1, max_length)
    out = torch.full((B * k, L), pad, dtype=torch.long)
    for i, h in enumerate(best):
        out[i, :len(h)] = torch.tensor(h)
        if len(h) < max_length:
            out[i, len(h)] = eos
    return out.to(device), torch.tensor(best_scores)


def beam_search_device(step_fn, reorder_fn, B: int, k: int, V: int, max_length: int, bos: int, eos: int, pad: int,
                       device, stop_rule: str = "hf4", sync_every: int = 8) -> Tuple[torch.Tensor, torch.Tensor]:
    """The same search with the bookkeeping on the device (afm_beam_step / afm_beam_finalize): log-softmax, top-2k,
    hypothesis pool, running sequences and beam reorder indices never leave HBM; the host reads one int32 (number of
    open samples) every `sync_every` tokens to stop early.  step_fn(last ids (B*k,) int64) -> fp32 logits (B*k, V);
    reorder_fn(beam_idx (B*k,) int32 device tensor)."""
    from . import ops
    i32 = dict(dtype=torch.int32, device=device)
    seq = [torch.full((B * k, max_length), pad, dtype=torch.long, device=device) for _ in range(2)]
    seq[0][:, 0] = bos
    beam_scores = torch.zeros(B, k, dtype=torch.float32, device=device)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1).contiguous()
    beam_idx = torch.zeros(B * k, **i32)
    hyp_seq = torch.zeros(B * k, max_length, dtype=torch.long, device=device)
    hyp_score = torch.zeros(B * k, dtype=torch.float32, device=device)
    hyp_len, hyp_count, done, n_open = torch.zeros(B * k, **i32), torch.zeros(B, **i32), torch.zeros(B, **i32), torch.zeros(1, **i32)
    rule = {"hf4": 0, "hf5": 1}[stop_rule]
    cur, cur_len = 0, 1

    def desc(logits, length):
        return ops.beam_desc(B, k, V, length, max_length, eos, pad, rule, logits, seq[cur], seq[cur ^ 1], beam_scores, beam_idx,
                             hyp_seq, hyp_score, hyp_len, hyp_count, done, n_open)
    while cur_len < max_length:
        logits = step_fn(seq[cur][:, cur_len - 1])
        ops.beam_step(desc(logits.float() if logits.dtype != torch.float32 else logits, cur_len))
        cur ^= 1
        cur_len += 1
        if cur_len < max_length:
            reorder_fn(beam_idx)
        if cur_len % sync_every == 0 and int(n_open.item()) == 0:
            break
    out = torch.empty(B * k, max_length, dtype=torch.long, device=device)
    out_scores = torch.empty(B * k, dtype=torch.float32, device=device)
    out_len = torch.empty(B * k, **i32)
    ops.beam_finalize(desc(None, cur_len), out, out_scores, out_len)
    L = int(out_len.max().item())
    return out[:, :L], out_scores
