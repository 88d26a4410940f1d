"""CPU: host-side logic of the product (schedule, step count, parameter layout, synthetic batches)."""
import numpy as np
import torch

from tests import golden_io as G


def test_onecycle_matches_torch_golden():
    from multimodalanalytical_amd.optim import onecycle
    t = G.load("schedule")
    for total in (10, 100):
        lr = [onecycle(s, total, 1e-3)[0] for s in range(total)]
        b1 = [onecycle(s, total, 1e-3)[1] for s in range(total)]
        np.testing.assert_allclose(lr, t[f"onecycle{total}"]["lr"].numpy(), rtol=1e-12)
        np.testing.assert_allclose(b1, t[f"onecycle{total}"]["beta1"].numpy(), rtol=1e-12)
    import pytest
    with pytest.raises(ValueError):
        onecycle(10, 10, 1e-3)      # torch raises when stepped past total_steps


def test_calculate_training_steps():
    from multimodalanalytical_amd.trainer import calculate_training_steps
    assert calculate_training_steps(177000, 128, 4, 60) == 346 * 60          # reference utils.py:156-172
    assert calculate_training_steps(20, 128, 4, 1) == 1
    assert calculate_training_steps(177000, 128, 4, 60, world_size=8) == 44 * 60


def test_param_layout_equals_reference_state_dict():
    from multimodalanalytical_amd.params import ParamStore, build_specs
    for name in ("model_plain", "model_gated_learned"):
        t = G.load(name)
        cfg = G.model_cfg(t["meta"])
        specs = build_specs(cfg, t["meta"]["data_config"], 26)
        ps = ParamStore(specs, torch.device("cpu"), with_bf16=False)
        ref = {k: v for k, v in t["sd"].items() if not k.endswith("positional_encodings.pos_enc")}
        assert set(ps.specs) == set(ref)
        for k, v in ref.items():
            assert ps.specs[k].shape == tuple(v.shape), k
            assert ps.specs[k].offset % 8 == 0
        ps.load(ref)
        for k, v in ref.items():
            assert torch.equal(ps.p(k), v)
        if cfg["gated_linear"]:   # linear1 | gate must be contiguous for the fused (2f x d) GEMM
            a, b = ps.specs["encoder.layers.0.linear1.weight"], ps.specs["encoder.layers.0.gate.weight"]
            assert a.offset + a.numel == b.offset
        ps.init_(1)
        w = ps.p("token_ff.weight")
        bound = (6.0 / (w.shape[0] + w.shape[1])) ** 0.5
        assert float(w.abs().max()) <= bound and float(w.abs().max()) > 0.5 * bound   # xavier_uniform_


def test_sincos_table_matches_reference_golden():
    from multimodalanalytical_amd.engine import sincos_table
    t = G.load("schedule")
    for d in (64, 128, 30):
        torch.testing.assert_close(sincos_table(d, 40), t["sincos"][str(d)], rtol=0, atol=1e-6)


def test_synthetic_workloads_follow_the_collator_contract():
    from multimodalanalytical_amd import synth
    for name, S in (("c1", 26), ("c2", 1024), ("c3", 1024), ("c4", 1024), ("c5", 56)):
        b, w = synth.make_batch(name, 3, seed=5)
        assert b["encoder_pad_mask"].shape == (S, 3) and b["encoder_pad_mask"].dtype == torch.bool
        T = w["T"]
        assert b["target"].shape == (T, 3) and b["decoder_input"]["Smiles"].shape == (T, 3)
        assert torch.equal(b["decoder_input"]["Smiles"][1:], b["target"][:-1])       # shifted by one
        assert (b["decoder_input"]["Smiles"][0] == synth.BOS).all()
        assert torch.equal(b["decoder_pad_mask"], b["decoder_input"]["Smiles"] == synth.PAD)
        b2, _ = synth.make_batch(name, 3, seed=5)
        assert torch.equal(b["target"], b2["target"])                                  # seeded
    assert abs(synth.train_flops_per_sample(synth.WORKLOADS["c2"]["cfg"], 1024, 128, 128) / 1e9 - 196.3) < 0.1
    assert abs(synth.train_flops_per_sample(synth.WORKLOADS["c4"]["cfg"], 1024, 128, 128) / 1e9 - 1013) < 1


def test_dropmask_helpers_are_consistent():
    from tests.dropmask import keep_mask, keep_mask16
    k = keep_mask(0.1, 123, 4, 200000)
    assert abs(k.mean() - 0.9) < 0.005
    k16, scale = keep_mask16(0.1, 123, 4, 200000, 200)
    assert abs(k16.mean() - 0.9) < 0.005 and abs(scale - 1 / 0.9) < 1e-3
    assert keep_mask(0.0, 1, 1, 10).all()


def test_mixture_index_stream_matches_oracle_restatement():
    """preprocess.mix_indices (host side of the mixture generator) == the oracle's restatement of
    data/datasets.py:59-116 for the same seed; both follow numpy's global RNG as the reference does."""
    import numpy as np
    from multimodalanalytical_amd.preprocess import mix_indices
    from oracle import afm_oracle as O
    cfg = dict(n_compounds=2, compounds_ratio=None, parallel_samples=16, train_max_n_samples=64, normalize=True)
    a = list(mix_indices(30, cfg, "train", seed=3247))
    b = list(O.mix_indices(30, cfg, "train", seed=3247))
    assert len(a) == len(b) == 4 and all(np.array_equal(x, y) for x, y in zip(a, b))
    assert all((r[:, 0] != r[:, 1]).all() for r in a)
    cfg3 = dict(n_compounds=3, parallel_samples=100, train_max_n_samples=10, normalize=False)
    assert [len(x) for x in mix_indices(5, cfg3, "train")] == [len(x) for x in O.mix_indices(5, cfg3, "train")]


def test_interleave_rounds_is_zip_longest_over_the_record_streams():
    """preprocess.interleave_rounds == `multi_config_mix` (data/datasets.py:24-46): the record streams of several mixture
    configurations alternate record by record, an exhausted stream drops out."""
    from itertools import zip_longest
    from multimodalanalytical_amd.preprocess import interleave_rounds

    def stream(tag, sizes):
        base = 0
        for n in sizes:
            yield {"indices": None, "IR": torch.arange(base, base + n).float()[:, None] + 1000.0 * tag,
                   "compound": torch.arange(base, base + n) + 1000 * tag}
            base += n
    layouts = [[(0, [5, 3, 4])], [(0, [5, 3]), (1, [2, 9, 1])], [(0, [4]), (1, [4, 4]), (2, [1, 1, 1, 7])]]
    for lay in layouts:
        got = torch.cat([r["compound"] for r in interleave_rounds([stream(t, sz) for t, sz in lay])]).tolist()
        flat = [[1000 * t + i for i in range(sum(sz))] for t, sz in lay]
        want = [x for row in zip_longest(*flat, fillvalue=None) for x in row if x is not None]
        assert got == want, lay
        ir = torch.cat([r["IR"] for r in interleave_rounds([stream(t, sz) for t, sz in lay])])[:, 0].long().tolist()
        assert ir == want


def test_param_spans_must_be_contiguous():
    """linear1|gate and the packed biases are read as ONE view: dimensions that leave alignment padding between
    the tensors must be refused, not silently shifted (params.ParamStore.span / vec_span)."""
    import pytest
    from multimodalanalytical_amd import synth
    from multimodalanalytical_amd.params import ParamStore, build_specs
    wl = synth.WORKLOADS["c5"]
    ps = ParamStore(build_specs(wl["cfg"], wl["data"], 128), "cpu", False)
    f, d = wl["cfg"]["encoder_ffn_dim"], wl["cfg"]["d_model"]
    assert ps.span(ps.flat, "encoder.layers.0.linear1.weight", 2 * f, d).shape == (2 * f, d)
    assert ps.vec_span(ps.flat, "encoder.layers.0.linear1.bias", 0, 2 * f).numel() == 2 * f
    bad = dict(wl["cfg"], d_model=36, encoder_ffn_dim=10, decoder_ffn_dim=10, encoder_attention_heads=4,
               decoder_attention_heads=4)
    ps = ParamStore(build_specs(bad, wl["data"], 128), "cpu", False)
    with pytest.raises(ValueError):
        ps.vec_span(ps.flat, "encoder.layers.0.linear1.bias", 0, 20)


def test_elementwise_dropout_stream_statistics():
    """The two-level element-wise dropout stream (csrc/afm_common.h afm_keep / afm_keep_scale, restated in tests/dropmask.py): keep
    rate within 3 sigma of 1 - p and serial correlations at lags 1-4, 16, 64 (the block length) inside 4.5 sigma (lag 2 and
    lag 16 sit at -2 .. -3 sigma, i.e. about -1e-3, on every seed; tools/hash_quality.py prints the table)."""
    import numpy as np
    from tests.dropmask import keep_mask
    n = 1 << 21
    for seed, site in ((12345, 7), (99, 3)):
        k = keep_mask(0.1, seed, site, n).astype(np.float64)
        assert abs(k.mean() - 0.9) < 3 * np.sqrt(0.09 / n) + 2.0 ** -16
        s = k - k.mean()
        for lag in (1, 2, 3, 4, 16, 64):
            c = float((s[:-lag] * s[lag:]).mean() / s.var()) * np.sqrt(n)
            assert abs(c) < 4.5, (seed, site, lag, c)


def test_validation_pass_counts_the_trailing_short_batch():
    """a mixture validation split whose nominal length is not a multiple of the batch size has a trailing short batch,
    which len() counts and the validation pass visits; a split without a single batch raises before the loop (the reference's
    loaders all run with drop_last=False, data/datamodules.py:433,467,502)."""
    import pytest
    from multimodalanalytical_amd.cli import training as T

    def loader(nominal, bs, key, world=1):
        ld = T.MixtureLoader.__new__(T.MixtureLoader)
        ld.nominal, ld.bs, ld.world, ld.key = nominal, bs, world, key
        n_batches = -(-(nominal // world) // bs) if key != "train" else (nominal // world) // bs
        ld.epoch = lambda epoch: iter(range(n_batches))          # what epoch() yields: full batches, then (not for train) a short one
        return ld

    assert len(loader(100, 16, "validation")) == 7 and len(loader(100, 16, "train")) == 6 and len(loader(96, 16, "test")) == 6
    assert len(loader(5, 16, "validation")) == 1 and len(loader(5, 16, "train")) == 0 and len(loader(100, 16, "validation", world=2)) == 4
    seen = [i for i, _ in T.val_batches(loader(100, 16, "validation"), 1.0)]
    assert seen == list(range(7))                                  # the short batch (index 6) is validated
    assert [i for i, _ in T.val_batches(loader(5, 16, "validation"), 1.0)] == [0]      # nominal < batch size: one short batch
    assert [i for i, _ in T.val_batches(loader(100, 16, "validation"), 3)] == [0, 1, 2]
    assert [i for i, _ in T.val_batches(loader(100, 16, "validation"), 0.5)] == [0, 1, 2]
    with pytest.raises(RuntimeError, match="no batch"):
        list(T.val_batches(loader(0, 16, "validation"), 1.0))


def test_profile_tools_on_a_synthetic_trace(tmp_path):
    """tools/rocpd_stats.py --by-grid --cluster and tools/rocpd_overlap.py --standin on a hand-made rocpd database: rows split by grid and by
    duration band; the exchange's stand-in copies are found on the side queue and matched with the compute kernels that run meanwhile."""
    import os
    import sqlite3
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    db = tmp_path / "t.db"
    con = sqlite3.connect(db)
    con.execute("create table rocpd_info_kernel_symbol (id integer, kernel_name text)")
    con.execute("create table rocpd_kernel_dispatch (kernel_id integer, start integer, end integer, queue_id integer, grid_size_x integer, "
                "grid_size_y integer, grid_size_z integer, workgroup_size_x integer, group_segment_size integer)")
    names = {1: "k_gemm", 2: "k_attn", 3: "__amd_rocclr_copyBuffer.kd", 4: "k_adam", 5: "k_sumsq", 6: "k_patch_x"}
    for i, n in names.items():
        con.execute("insert into rocpd_info_kernel_symbol values (?, ?)", (i, n))
    t = 1000
    rows = []
    for step in range(2):
        rows.append((6, t, t + 10, 4, 256, 1, 1, 256, 0)); t += 20
        for layer in range(4):
            rows.append((1, t, t + 100_000, 4, 65536, 1, 1, 256, 1000)); t += 100_010          # persistent GEMM: one grid, two bands
            rows.append((1, t, t + 300_000, 4, 65536, 1, 1, 256, 1000)); t += 300_010
            rows.append((2, t, t + 50_000, 4, 8192 * 256 if layer % 2 else 1024 * 256, 1, 1, 256, 500)); t += 50_010
            rows.append((3, t - 40_000, t - 30_000, 1, 1024, 1, 1, 256, 0))                      # a bucket's stand-in copy on queue 1, beside the attention kernel
        rows.append((3, t + 5, t + 4000, 1, 1024, 1, 1, 256, 0))                                 # the last bucket: after the backward
        rows.append((5, t + 4100, t + 4200, 4, 64, 1, 1, 256, 0))
        rows.append((4, t + 5000, t + 6000, 4, 4096 * 256, 1, 1, 256, 0)); t += 7000
    con.executemany("insert into rocpd_kernel_dispatch values (?,?,?,?,?,?,?,?,?)", rows)
    con.commit(); con.close()
    tools = os.path.join(ROOT, "tools")
    out = subprocess.run([sys.executable, os.path.join(tools, "rocpd_stats.py"), str(db), "--by-grid", "--cluster", "1.3"], capture_output=True, text=True, check=True).stdout
    lines = [ln.lstrip('"') for ln in out.splitlines() if ln.lstrip('"').startswith("k_")]
    assert sum("k_gemm" in ln and "#band0" in ln for ln in lines) == 1 and sum("k_gemm" in ln and "#band1" in ln for ln in lines) == 1
    assert sum("k_attn [grid 8192 x 256" in ln for ln in lines) == 1 and sum("k_attn [grid 1024 x 256" in ln for ln in lines) == 1
    ov = subprocess.run([sys.executable, os.path.join(tools, "rocpd_overlap.py"), str(db), "--standin"], capture_output=True, text=True, check=True).stdout
    assert "5 RCCL kernels on queue(s) [1], compute on queue(s) [4]" in ov
    assert ov.count("k_attn x1") == 4 and "0 kernels: -" in ov            # four buckets beside backward kernels, the last one alone


def test_isa_spill_check_flags_a_spill_of_an_inflight_asm_read(tmp_path):
    """tools/isa_asm_spill_check.py on hand-made ISA: a scratch store of a register an asm `ds_read_b64_tr_b16` wrote, with no
    `s_waitcnt lgkmcnt(0)` in between, is a hazard (the compiler does not know the read is asynchronous); behind the wait it is not."""
    import os
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from isa_asm_spill_check import check
    bad = tmp_path / "bad.s"
    bad.write_text("_Z5k_badv: ; @k\n\tds_read_b64_tr_b16 v[46:47], v35 offset:0\n\tds_read_b64_tr_b16 v[48:49], v35 offset:1024\n"
                   "\tscratch_store_dwordx4 off, v[46:49], off ; 16-byte Folded Spill\n\ts_endpgm\n.Lfunc_end0:\n")
    good = tmp_path / "good.s"
    good.write_text("_Z6k_goodv: ; @k\n\tds_read_b64_tr_b16 v[46:47], v35 offset:0\n\ts_waitcnt lgkmcnt(0)\n"
                    "\tscratch_store_dwordx2 off, v[46:47], off ; 8-byte Folded Spill\n\tds_read_b64_tr_b16 v[10:11], v35 offset:0\n"
                    "\tscratch_store_dword off, v12, off offset:8\n\ts_endpgm\n.Lfunc_end0:\n")
    assert check(str(bad)) == 1
    assert check(str(good)) == 0
