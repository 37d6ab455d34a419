#!/usr/bin/env python3
"""Benchmark of the trim + pileup + call hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--depth D]

One "step" = one pass of the whole hot path over one synthetic batch that is already resident
in HBM: reset the device table, run the trim + pileup kernels over every read of the batch
(trimmed CIGAR / position / flags written back to HBM), [N > 1: ONE all-reduce of the count
tables over RCCL], then call every reference position (device k_call / k_call_compact + host
record assembly and the consensus string; with N > 1 every rank holds the reduced table and makes
the same calls, so no second collective is needed).  Two steps are in flight (two engines on two
HIP streams): the host side of step k-1 and its all-reduce overlap the kernels of step k; every
step is complete when the timed region ends.  --no-pipeline runs them strictly one at a time.

Workload (config.workload): BASELINE.json's metric is quoted on a ~30 kb reference at 10k x
depth: 29,903 nt synthetic genome, ARTIC-style 98-amplicon primer scheme, 150 bp paired reads,
1,993,533 reads per GPU (SURVEY.md section 8(d) generator, built on the GPU with torch).  With
N GPUs the genome's amplicons are range-partitioned by coordinate over the ranks and every rank
holds a full-size batch of its own range (weak scaling: the job is N x 10k x deep); counts are
stitched with one reduce of the 7 x 29,903 uint32 device table.

Prints ONE JSON line on rank 0 (see the driver contract), including
  roofline:     algorithmic bytes of the CIGAR-scan kernel / its HIP-event duration vs 8 TB/s
  cpu_baseline: the C restatement in oracle/ timed on this host's cores on the same batch
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def main():
    # RCCL prints a banner on stdout when a communicator is created; the contract is ONE JSON line on
    # stdout, so everything else written to fd 1 (from any library) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--depth", type=int, default=10000, help="mean coverage per GPU (10000 -> 1,993,533 reads)")
    ap.add_argument("--cpu-passes", type=int, default=12, help="passes of the CPU baseline over the batch (0 = skip)")
    ap.add_argument("--variant", type=int, default=4)
    ap.add_argument("--no-pipeline", action="store_true", help="one step at a time (default: two steps in flight)")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-rank code path (RCCL all-reduce) even with one rank")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: amplipy_amd has no CPU execution path")
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    assert world == args.gpus or world == 1 and args.gpus == 1, "launch with torch.distributed.run for --gpus > 1"

    from amplipy_amd import abi, calling, lib, parallel, synth, synth_torch

    genome = synth.make_genome()
    primers, amps = synth.make_artic_scheme()
    G = int(genome.size)
    ref_seq = synth.genome_string(genome)
    n_reads = synth.reads_for_depth(args.depth)
    # range partition by coordinate: rank r owns a contiguous run of amplicons
    a_lo, a_hi = parallel.amplicon_range(amps.shape[0], rank, world)
    t_gen = time.time()
    batch = synth_torch.make_amplicon_batch_device(genome, amps[a_lo:a_hi], n_reads, seed=1000 + rank, device=dev)
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen

    mn, mx, mpl = lib.find_overlapping_primers(G, [(s, e) for s, e, _ in primers], 0)
    rd = batch.struct()
    cp = calling.call_params(10, 0.0, 1, 0.03, True, True)

    class Slot:
        """One in-flight step: its own HIP stream, engine (device table, event list, scratch) and outputs."""

        def __init__(self):
            self.stream = torch.cuda.Stream(device=dev)
            self.eng = eng = lib.Engine(G, device=local_rank)
            eng.set_kernel_variant(args.variant)
            eng.set_stream(self.stream.cuda_stream)
            self.table = torch.zeros(G * 7, dtype=torch.int32, device=dev)   # counts [G][6] + insertion tally [G]
            eng.bind_counts(self.table.data_ptr())
            eng.set_primers(mn, mx, mpl)
            eng.set_params(20, 4, True, True)
            eng.set_reference(ref_seq)
            eng.reserve_events(max(1 << 20, n_reads // 4))
            self.out = {
                "new_pos": torch.zeros(n_reads, dtype=torch.int32, device=dev),
                "new_ncig": torch.zeros(n_reads, dtype=torch.int32, device=dev),
                "new_cig": torch.zeros(batch.n_cig + 3 * n_reads, dtype=torch.int32, device=dev),
                "ref_len": torch.zeros(n_reads, dtype=torch.int32, device=dev),
                "trim_flags": torch.zeros(n_reads, dtype=torch.uint8, device=dev),
                "status": torch.zeros(n_reads, dtype=torch.uint8, device=dev),
            }
            self.dev_out = abi.AmpTrimOut(*[self.out[k].data_ptr() for k in
                                            ("new_pos", "new_ncig", "new_cig", "ref_len", "trim_flags", "status")])

        def ins_provider(self, positions):
            eng = self.eng
            ev = eng.events()
            keep = np.isin(ev["ref_pos"], np.fromiter(positions, np.int64, len(positions)))
            strings = eng.event_strings_device(rd, ev[keep]) if keep.any() else []
            pairs = list(zip(ev["ref_pos"][keep].tolist(), strings))
            if dist is not None:      # every rank sees the same positions (tables are all-reduced): symmetric exchange
                pairs = parallel.allgather_relevant_events(dist, world, pairs)
            return calling.tallies_from_events(pairs, positions)

    # Steps are software-pipelined `depth` deep: the kernels of step k run while the host assembles the
    # records of step k-1 (and, N > 1, while RCCL all-reduces its table).  Every step is complete -
    # kernels, reduce, calls, records, consensus string - before the timed region ends.
    depth = 1 if args.no_pipeline else 2
    slots = [Slot() for _ in range(depth)]
    scan_ms = []
    last = {}

    def submit(k):
        sl = slots[k % depth]
        with torch.cuda.stream(sl.stream):
            sl.eng.reset()
            sl.eng.process_device(rd, 0, sl.dev_out)
            if dist is not None:
                # ONE collective per step: afterwards every rank holds the whole job's table and makes the same calls
                parallel.allreduce_table(dist, sl.table)

    def finish(k):
        sl = slots[k % depth]
        with torch.cuda.stream(sl.stream):
            last["call"] = res = calling.call(sl.eng, ref_seq, cp, sl.ins_provider)
            last["consensus"] = res.consensus_string("N")
            last["slot"] = sl
        scan_ms.append(sl.eng.last_kernel_ms()[1])

    def run(n):
        for k in range(n):
            submit(k)
            if k >= depth - 1:
                finish(k - (depth - 1))
        for k in range(max(n - (depth - 1), 0), n):
            finish(k)

    run(args.warmup)
    del scan_ms[:]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    eng, table, out_t = last["slot"].eng, last["slot"].table, last["slot"].out
    # outside the timed region: the same kernel with nothing else on the GPU (the pipelined steps above
    # overlap it with the previous step's small kernels and with the next scan, which stretches its
    # own duration while shortening the step)
    solo_ms = []
    if depth > 1:
        for _ in range(5):
            with torch.cuda.stream(last["slot"].stream):
                eng.reset()
                eng.process_device(rd, 0, last["slot"].dev_out)
                eng.sync()
            solo_ms.append(eng.last_kernel_ms()[1])

    # ---- accounting (outside the timed region) ------------------------------------------------
    n_out = int(out_t["new_ncig"].sum().item())
    n_err = int((out_t["status"] != 0).sum().item())
    L = 150
    alg_bytes = n_reads * (16 + (L + 1) // 2 + L + 8) + 4 * batch.n_cig + 4 * n_out + 6 * G * 4 + 2 * G * 4
    k_ms = float(np.mean(scan_ms))
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    traffic = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.isfile(tf):
        traffic = json.load(open(tf)).get("depth%d" % args.depth)

    cpu = None
    if rank == 0 and world == 1 and args.cpu_passes > 0:
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle
        hb = batch.to_host()
        # one thread: the checker pass (the GPU result of the last step must equal the oracle's on the same batch)
        t1 = time.perf_counter()
        ref = oracle.process(hb, G, mn, mx, mpl, 20, 4)
        t1 = time.perf_counter() - t1
        got = table.cpu().numpy().view(np.uint32)
        assert np.array_equal(got[:G * 6].reshape(G, 6), ref.counts), "GPU count table differs from the oracle"
        assert np.array_equal(out_t["new_pos"].cpu().numpy(), ref.trim.new_pos), "trimmed positions differ from the oracle"
        assert int(got[G * 6:].sum()) == ref.events.size, "insertion events differ from the oracle"
        # all host cores: reads sharded over threads (the C call releases the GIL), private tables summed
        # (a one-GPU box gets a 16-CPU share of its host, whatever the affinity mask says)
        cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("AMPLIPY_CPU_THREADS", "16")))
        cuts = [hb.n * k // cores for k in range(cores + 1)]

        def shard(k):
            return oracle.process(hb, G, mn, mx, mpl, 20, 4, lo=cuts[k], hi=cuts[k + 1]).counts

        with ThreadPoolExecutor(cores) as pool:
            t_cpu = time.perf_counter()
            for _ in range(args.cpu_passes):
                parts = list(pool.map(shard, range(cores)))
                total = np.sum(parts, axis=0, dtype=np.uint32)
            t_cpu = time.perf_counter() - t_cpu
        assert np.array_equal(total, ref.counts), "sharded CPU run differs from the single-thread run"
        cpu = {"value": round(hb.n * args.cpu_passes / t_cpu, 1), "unit": "reads/s", "cores": cores, "kind": "port",
               "single_thread_value": round(hb.n / t1, 1),
               "sample": "the full %d-read batch, %d passes of oracle/amplipy_oracle.c (trim + pileup) sharded by read over "
                         "%d threads with private count tables summed at the end; single_thread_value = one pass on one thread"
                         % (hb.n, args.cpu_passes, cores)}

    if rank == 0:
        total_reads = n_reads * world * args.steps
        res = last["call"]
        line = {
            "metric": "aligned reads/s through trim+pileup+call, 30 kb ref @ 10k x depth",
            "value": round(total_reads / elapsed, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32", "data": "synthetic",
            "config": {"workload": "synthetic 29,903 nt genome, 98-amplicon ARTIC-style primers, 150 bp paired reads, "
                                   "%d x depth = %d reads per GPU, inputs resident in HBM" % (args.depth, n_reads),
                       "reads_per_gpu": n_reads, "read_len": L, "ref_len": G, "min_quality": 20, "window": 4,
                       "parallelism": "coordinate-range partition x%d + one RCCL all-reduce of the count table per step" % world,
                       "steps_in_flight": depth,
                       "kernel_variant": args.variant, "error_reads": n_err,
                       "variants_called": res.n_records, "ins_relevant_positions": res.n_relevant},
            "roofline": {"bound": "hbm", "kernel": "k_tile" if args.variant == 2 else "k_reads_lane",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(k_ms, 5),
                         "kernel_ms_alone": round(float(np.mean(solo_ms)), 5) if solo_ms else round(k_ms, 5),
                         "frac_alone": round(alg_bytes / (float(np.mean(solo_ms)) if solo_ms else k_ms) / 1e6 / HBM_PEAK_GBS, 5),
                         "note": "kernel_ms / achieved / frac: HIP events around k_tile inside the timed region, where the "
                                 "launches of two steps in flight overlap each other and the small kernels; "
                                 "kernel_ms_alone / frac_alone: the same launch with the GPU to itself (5 launches after the timed region)"},
            "cpu_baseline": cpu,
            "gen_seconds": round(t_gen, 2),
        }
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    for sl in slots:
        sl.eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
