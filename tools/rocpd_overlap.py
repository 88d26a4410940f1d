"""Gradient exchange against backward in a rocprofv3 rocpd database of `bench.py --force-ddp`:

    python tools/rocpd_overlap.py results.db > profiles/r06_c2_ddp_timeline.txt

For the LAST optimiser step of the trace: every RCCL kernel (the buckets of afm_allreduce_bucket on the reducer's side stream) with its
start relative to the step's last micro-batch, its duration, the queue it ran on, and the compute kernels that were executing while it
ran; then the gap between the end of the last backward kernel and the start of k_adam, and how much of the exchange lay outside the
backward pass.  With one rank the all-reduce moves no bytes between GPUs: what the trace shows is WHERE the buckets are launched and
that they run beside the backward kernels of earlier layers, not what xGMI would make of them.
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name).replace("void ", "").strip()
    m = re.match(r"_ZN7afm_(?:f16|bf16)\d+([a-z_0-9]+?)I", name) or re.match(r"_Z\d+([a-z_0-9A-Z]+?)(?:I|E|P)", name)
    return (m.group(1) if m else name)[:44]


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = {r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)").fetchall()}
    qcol = "d.queue_id" if "queue_id" in cols else ("d.stream_id" if "stream_id" in cols else "0")
    rows = db.execute(f"select s.kernel_name, d.start, d.end, {qcol} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                      "on d.kernel_id = s.id order by d.start").fetchall()
    is_rccl = lambda n: "nccl" in n.lower() or "rccl" in n.lower()
    standin = "--standin" in sys.argv
    if standin:
        # one rank: RCCL enqueues nothing for an in-place all-reduce, the reducer's AFM_DDP_STANDIN copy of each bucket stands in for it --
        # the device-copy kernels on a queue other than the compute queue (the one with the most dispatches)
        counts = {}
        for r in rows:
            counts[r[3]] = counts.get(r[3], 0) + 1
        main_q = max(counts, key=counts.get)
        is_rccl = lambda n, q=None: False
        rows_tagged = [(("STANDIN " + n) if ("copyBuffer" in n and q != main_q and en - st > 3000) else n, st, en, q) for n, st, en, q in rows]
        rows = rows_tagged
        is_rccl = lambda n: n.startswith("STANDIN ")
    adam = [r for r in rows if "k_adam" in r[0]]
    if not adam:
        raise SystemExit("no k_adam dispatch in the trace")
    a_start = adam[-1][1]
    prev_adam_end = adam[-2][2] if len(adam) > 1 else rows[0][1]
    step = [r for r in rows if prev_adam_end <= r[1] <= a_start]
    rccl = [r for r in step if is_rccl(r[0])]
    comp = [r for r in step if not is_rccl(r[0])]
    if not rccl:
        raise SystemExit("no RCCL kernel between the last two optimiser steps (one rank? RCCL enqueues nothing then: trace with AFM_DDP_STANDIN=1 "
                         "and pass --standin)")
    # the backward of the last micro-batch ends with the last weight-gradient launch / embedding backward before the clip (k_sumsq)
    sumsq = [r for r in comp if "k_sumsq" in r[0]]
    bwd_end = max(r[2] for r in comp if r[1] < (sumsq[0][1] if sumsq else a_start))
    t0 = rccl[0][1]
    if standin:
        print("# ONE rank: the kernels below are the reducer's stand-in copies of each bucket (AFM_DDP_STANDIN=1), launched where the RCCL "
              "all-reduce of that bucket is launched; RCCL itself enqueues nothing for one rank")
    print(f"# optimiser step ending at k_adam: {len(step)} dispatches over {(a_start - prev_adam_end) / 1e6:.2f} ms; "
          f"{len(rccl)} RCCL kernels on queue(s) {sorted({r[3] for r in rccl})}, compute on queue(s) {sorted({r[3] for r in comp})}")
    print(f"# first bucket starts {(bwd_end - t0) / 1e3:.1f} us BEFORE the last backward kernel ends")
    print(f"{'bucket':>6s} {'start us (rel. first bucket)':>30s} {'dur us':>9s} {'queue':>6s}  compute kernels running meanwhile")
    outside = 0
    for i, (n, st, en, q) in enumerate(rccl):
        over = [c for c in comp if c[1] < en and c[2] > st]
        names = {}
        for c in over:
            names[short(c[0])] = names.get(short(c[0]), 0) + 1
        desc = ", ".join(f"{k} x{v}" for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:5]) or "-"
        outside += max(0, en - max(st, bwd_end))
        print(f"{i:6d} {(st - t0) / 1e3:30.1f} {(en - st) / 1e3:9.1f} {q!s:>6s}  {len(over)} kernels: {desc}")
    last_rccl_end = max(r[2] for r in rccl)
    print(f"# last backward kernel ends -> k_adam starts: {(a_start - bwd_end) / 1e3:.1f} us "
          f"(clip's k_sumsq and the scalar kernels in between); last RCCL kernel ends {(last_rccl_end - bwd_end) / 1e3:.1f} us after the backward")
    print(f"# RCCL time outside the backward pass: {outside / 1e3:.1f} us of {sum(r[2] - r[1] for r in rccl) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
