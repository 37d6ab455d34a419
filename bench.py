#!/usr/bin/env python3
"""Benchmark of the trim + pileup + call hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--depth D] [--strong]

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
1,993,533 reads per GPU (SURVEY.md section 8(d) generator, built on the GPU with torch).
  default (weak scaling): with N GPUs the genome's amplicons are range-partitioned by coordinate
      over the ranks and every rank holds a full-size batch of its own range (the job is N x D deep)
  --strong (BASELINE config 4): ONE job of depth D; its coordinate-sorted reads are cut into N
      contiguous slices of equal base counts (parallel.shard_bounds), rank r processes slice r, and
      rank 0 asserts that the all-reduced table is bit-identical to the table of the whole job
      processed on one GPU.
Either way the counts are stitched with one all-reduce of the 7 x 29,903 uint32 device table.

Prints ONE JSON line on rank 0 (see the driver contract), including
  roofline:     algorithmic bytes of the CIGAR-scan pass / its HIP-event duration vs 8 TB/s.  The
                scan pass is what amp_process_batch_device launches: k_fast (simple reads), k_tile<LIST>
                (the other reads, straight from the fast kernel's per-block lists), k_deferred_heavy
  cpu_baseline: the C restatement in oracle/ timed on this host's cores on the same batch, plus the
                pure-Python restatement on a 100 k-read subsample
  e2e:          file-to-file rates outside the timed region (host pointers, BAM -> calls, BAM -> BAM)
  extra:        BASELINE configs 3 (100k x depth, 19.9 M reads) and 5 (8.0 M mixed 75-300 bp reads), each as ONE launch behind
                the timed region: duration of the scan pass alone, roofline fraction, oracle assertion
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
SCAN_KERNELS = {0: "scan pass: k_fast (k_fast5 for reads of more than 152 bases, k_fast7 for long reads with three CIGAR ops a read and more) + k_tile<LIST> + k_deferred_heavy",
                4: "scan pass: k_fast + k_tile<LIST> + k_deferred_heavy", 5: "scan pass: k_fast5 + k_tile<LIST> + k_deferred_heavy",
                6: "scan pass: k_fast6 + k_tile<LIST> + k_deferred_heavy", 7: "scan pass: k_fast7 + k_tile<LIST> + k_deferred_heavy",
                2: "scan pass: k_tile + k_deferred_heavy",
                3: "scan pass: k_trim + k_scan + k_tile<SPLIT> + k_deferred_heavy", 1: "k_reads_lane"}


def main():
    # RCCL prints a banner on stdout when a communicator is created; the contract is ONE JSON line on
    # stdout, so everything else written to fd 1 (from any library) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (200 x 0.23 ms: with eight steps in flight a 20-step region is two and a half pipeline depths long and its fill and drain cost 10 %: 0.248 ms per step at 20, 0.226 at 200, 0.223 at 1000)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--depth", type=int, default=10000, help="mean coverage (10000 -> 1,993,533 reads): per GPU, or of the whole job with --strong")
    ap.add_argument("--strong", action="store_true", help="BASELINE config 4: one job of --depth partitioned over the ranks (strong scaling)")
    ap.add_argument("--cpu-passes", type=int, default=0, help="passes of the CPU baseline over the batch (0 = as many as fit ~10 s; -1 = skip the CPU legs)")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (0 = chosen per batch by the library)")
    ap.add_argument("--no-pipeline", action="store_true", help="one step at a time (default: eight steps in flight over eight streams)")
    ap.add_argument("--in-flight", type=int, default=8, help="steps in flight (each has its own engine, table and outputs)")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-rank code path (RCCL all-reduce) even with one rank")
    ap.add_argument("--no-e2e", action="store_true", help="skip the file-to-file legs behind the timed region")
    ap.add_argument("--no-extra", action="store_true", help="skip the other BASELINE configs behind the timed region (config 3: 100k x depth; config 5: 8.0 M mixed reads)")
    ap.add_argument("--streams", type=int, default=8, help="HIP streams the steps in flight are dealt over (1: the GPU runs step after step; 2+: the next step's reads start on the CUs the last blocks of this step's reads have left)")
    ap.add_argument("--cu-share", type=int, default=0, help="size each step's scan pass for 1 / N of the CUs (0 = 2 when the steps are dealt over several streams, else 1)")
    ap.add_argument("--no-comm-overlap", action="store_true", help="multi-rank runs: all-reduce on the work stream, in front of the step's calls (default: on its own stream, under the next step's reads)")
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
    job_reads = synth.reads_for_depth(args.depth)
    t_gen = time.time()
    if args.strong:
        # the SAME job on every rank (same seed), cut by base count; rank r keeps rows [lo, hi)
        whole = synth_torch.make_amplicon_batch_device(genome, amps, job_reads, seed=1000, device=dev)
        cuts = parallel.shard_bounds_from_lseq(whole.lseq.cpu().numpy().view(np.uint32), world)
        lo, hi = cuts[rank], cuts[rank + 1]
        batch = whole.rows(lo, hi)
    else:
        # range partition by coordinate: rank r owns a contiguous run of amplicons and a full-size batch of them
        a_lo, a_hi = parallel.amplicon_range(amps.shape[0], rank, world)
        whole = None
        batch = synth_torch.make_amplicon_batch_device(genome, amps[a_lo:a_hi], job_reads, seed=1000 + rank, device=dev)
    n_reads = batch.n
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen

    mn, mx, mpl = lib.find_overlapping_primers(G, [(s, e) for s, e, _ in primers], 0)
    cp = calling.call_params(10, 0.0, 1, 0.03, True, True)
    # Every step in flight works on a batch of ITS OWN (same generator, another seed: 8 x 0.5 GB in HBM), so that passes that run
    # side by side do not stream the same half gigabyte through the 256 MB Infinity Cache.  (--strong: ONE job, every slot holds
    # the rank's slice of it.)
    slot_batches = {}

    def batch_for(lane):
        if args.strong or lane == 0:
            return batch
        if lane not in slot_batches:
            slot_batches[lane] = synth_torch.make_amplicon_batch_device(genome, amps[a_lo:a_hi], job_reads, seed=1000 + rank + 7919 * lane, device=dev)
        return slot_batches[lane]

    # The steps in flight are dealt over a few HIP streams (each step has its own engine, table and outputs).  The scan pass is
    # one block per CU for its whole duration, so on ONE stream the chip idles through every pass's last tile and through the
    # small calling kernels behind it (0.264 ms per step for a 0.234 ms pass); with the next step on another stream its blocks
    # take the CUs as they come free and the small kernels (reset 8, calls 40 / 32 VGPRs, no LDS to speak of) run beside them:
    # 0.2405 ms with four steps over four streams, 0.233 with eight over eight (the default).  Two steps in flight are not enough for this: the calls of step k then sit
    # behind the blocks of reads(k+1) while the host waits for them with nothing else queued.
    work_streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, 1 if args.no_pipeline else args.streams))]
    # ... and every pass is sized for HALF the CUs: two passes side by side finish sooner than one after the other on the whole
    # chip (a block works twice as long, so its start-up, its flush and the idle end of its last round of tiles weigh half as
    # much: 0.224 -> 0.212 ms per step).  The solo leg behind the timed region runs the pass alone and therefore on all CUs.
    cu_share = args.cu_share if args.cu_share > 0 else (2 if len(work_streams) > 1 else 1)
    work_stream = work_streams[0]
    # multi-rank runs: the step's one collective goes to a stream of its own.  xGMI moves the 0.8 MB table while the work
    # stream already runs the next step's reads (other slot, other table); the step's calls are queued behind those reads
    # and wait for the reduced table through an event.  One step of latency, no idle GPU around the collective.
    comm_stream = torch.cuda.Stream(device=dev) if (dist is not None and not args.no_comm_overlap and not args.no_pipeline and args.in_flight > 1) else None

    class Slot:
        """One in-flight step: its own engine (device table, event list, scratch, pinned result image) and outputs."""

        def __init__(self, b=None, lane=0):
            b = batch_for(lane) if b is None else b
            self.batch = b
            self.stream = work_streams[lane % len(work_streams)]
            self.eng = eng = lib.Engine(G, device=local_rank)
            eng.set_kernel_variant(args.variant)
            eng.set_cu_share(cu_share)
            eng.set_stream(self.stream.cuda_stream)
            self.table = torch.zeros(G * 7, dtype=torch.int32, device=dev)   # counts [G][6] + insertion tally [G]
            eng.bind_counts(self.table.data_ptr())
            eng.set_primers(mn, mx, mpl)
            eng.set_params(20, 4, True, True)
            eng.set_reference(ref_seq)
            eng.reserve_events(max(1 << 20, b.n // 4))
            self.rd = b.struct()
            self.out = {
                "new_pos": torch.zeros(b.n, dtype=torch.int32, device=dev),
                "new_ncig": torch.zeros(b.n, dtype=torch.int32, device=dev),
                "new_cig": torch.zeros(b.n_cig + 3 * b.n, dtype=torch.int32, device=dev),
                "ref_len": torch.zeros(b.n, dtype=torch.int32, device=dev),
                "trim_flags": torch.zeros(b.n, dtype=torch.uint8, device=dev),
                "status": torch.zeros(b.n, dtype=torch.uint8, device=dev),
            }
            self.dev_out = abi.AmpTrimOut(*[self.out[k].data_ptr() for k in
                                            ("new_pos", "new_ncig", "new_cig", "ref_len", "trim_flags", "status")])
            self.ev_reads, self.ev_reduced = torch.cuda.Event(), torch.cuda.Event()
            self.call_due = False        # reads and all-reduce are queued, the calls are not yet

        def ins_provider(self, positions):
            eng = self.eng
            ev = eng.events()
            keep = np.isin(ev["ref_pos"], np.fromiter(positions, np.int64, len(positions)))
            strings = eng.event_strings_device(self.rd, ev[keep]) if keep.any() else []
            pairs = list(zip(ev["ref_pos"][keep].tolist(), strings))
            if dist is not None:      # every rank sees the same positions (tables are all-reduced): symmetric exchange
                pairs = parallel.allgather_relevant_events(dist, world, pairs)
            return calling.tallies_from_events(pairs, positions)

    # Steps are software-pipelined `depth` deep: the kernels of step k run while the host assembles the
    # records of step k-1.  Every step is complete -
    # kernels, reduce, calls, records, consensus string - before the timed region ends.
    depth = 1 if args.no_pipeline else max(1, args.in_flight)
    if comm_stream is not None:
        depth += 1        # the calls of a step are queued one step later: one more step in flight keeps the host's record assembly under the GPU's work
    slots = [Slot(lane=k) for k in range(depth)]
    pass_ms, fast_ms = [], []
    step_trace = [] if os.environ.get("BENCH_STEP_TRACE") else None      # development aid: when every step of the timed region was complete
    last = {}

    def begin_calls(sl):
        if sl.call_due:
            with torch.cuda.stream(sl.stream):
                sl.stream.wait_event(sl.ev_reduced)
                sl.eng.call_compact_begin(cp)
            sl.call_due = False

    def submit(k):
        sl = slots[k % depth]
        with torch.cuda.stream(sl.stream):
            sl.eng.reset()
            sl.eng.process_device(sl.rd, 0, sl.dev_out)
            if comm_stream is None:
                if dist is not None:
                    # ONE collective per step: afterwards every rank holds the whole job's table and makes the same calls
                    parallel.allreduce_table(dist, sl.table)
                # the calling kernels go in right behind: queued in front of the next step's reads, not behind them
                sl.eng.call_compact_begin(cp)
            else:
                sl.ev_reads.record(sl.stream)
        if comm_stream is not None:
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(sl.ev_reads)
                parallel.allreduce_table(dist, sl.table)
                sl.ev_reduced.record(comm_stream)
            sl.call_due = True
            # the calls of the step before: behind this step's reads on the work stream, its all-reduce ran under them
            begin_calls(slots[(k - 1) % depth])

    def finish(k):
        sl = slots[k % depth]
        begin_calls(sl)                  # (the last steps of a run have no later submit to do it)
        with torch.cuda.stream(sl.stream):
            last["call"] = res = calling.call(sl.eng, ref_seq, cp, sl.ins_provider)
            last["consensus"] = res.consensus_string("N")
            last["slot"] = sl
        t, s = sl.eng.last_kernel_ms()
        pass_ms.append(t); fast_ms.append(s)
        if step_trace is not None:
            step_trace.append(time.perf_counter())

    def run(n):
        for k in range(n):
            submit(k)
            if k >= depth - 1:
                finish(k - (depth - 1))
        for k in range(max(n - (depth - 1), 0), n):
            finish(k)

    # every slot runs once before the timed region whatever W is (first use allocates: pinned result image, event list, scratch)
    run(max(args.warmup, depth))
    del pass_ms[:], fast_ms[:]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if step_trace is not None:
        del step_trace[:]
    run(args.steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if step_trace is not None and rank == 0:
        ts = [round((x - t0) * 1e3, 3) for x in step_trace]
        sys.stderr.write("step completion times [ms from the start of the timed region]: %s\nregion %.3f ms\n" % (ts, elapsed * 1e3))
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    eng, table, out_t = last["slot"].eng, last["slot"].table, last["slot"].out
    batch = last["slot"].batch           # (the checks and the byte counts below are those of the LAST step's own batch)
    rd = last["slot"].rd
    # outside the timed region: the same launches with nothing else on the GPU (the pipelined steps above
    # overlap them with the previous step's small kernels and with the next scan, which stretches their
    # own duration while shortening the step)
    solo_pass, solo_fast = [], []
    eng.set_cu_share(1)                  # alone, the pass has the whole chip
    for it in range(11):                 # three untimed passes on the new grid, five timed, three with the first kernel timed on its own
        eng.set_timing(it >= 8)          # (timing the first kernel of the pass on its own costs an idle gap)
        with torch.cuda.stream(last["slot"].stream):
            eng.reset()
            eng.process_device(rd, 0, last["slot"].dev_out)
            eng.sync()
        t, s = eng.last_kernel_ms()
        if it >= 3:
            (solo_pass if it < 8 else solo_fast).append(t if it < 8 else s)
    eng.set_timing(False)
    eng.set_cu_share(cu_share)
    if dist is not None:     # leave the engine with the reduced table again (the check below and the calls use it)
        with torch.cuda.stream(last["slot"].stream):
            parallel.allreduce_table(dist, table)
        torch.cuda.synchronize()

    # ---- the box's own stream ceiling (report against the spec peak AND a measured copy): 1 GiB copied device to device,
    # bytes read + bytes written over the HIP-event time of the copies
    peak_measured = None
    try:
        src_t = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
        dst_t = torch.empty_like(src_t)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            dst_t.copy_(src_t)
        ev0.record()
        for _ in range(10):
            dst_t.copy_(src_t)
        ev1.record()
        torch.cuda.synchronize()
        peak_measured = 2.0 * src_t.numel() * 4 * 10 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
        del src_t, dst_t
        torch.cuda.empty_cache()
    except Exception:
        peak_measured = None
    # ---- accounting (outside the timed region) ------------------------------------------------
    n_out = int(out_t["new_ncig"].sum().item())
    n_err = int((out_t["status"] != 0).sum().item())
    L = 150
    # 16 B of per-read fields + CIGAR words in and out + packed bases + qualities + 8 B of per-read
    # results, plus once per launch the count table written and the primer tables read
    n_cig_in = int(batch.cig_off32[-1].item() - batch.cig_off32[0].item())
    alg_bytes = n_reads * (16 + (L + 1) // 2 + L + 8) + 4 * n_cig_in + 4 * n_out + 6 * G * 4 + 2 * G * 4
    k_ms = float(np.mean(pass_ms))
    k_solo = float(np.mean(solo_pass))
    # one stream: the pass's own HIP-event duration inside the timed region; several: the region's pass rate (see roofline.note)
    k_rate_ms = k_ms if len(work_streams) == 1 else elapsed / args.steps * 1e3
    traffic = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.isfile(tf):
        traffic = json.load(open(tf)).get("depth%d_variant%d" % (args.depth, args.variant or 4))

    strong_check = None
    if args.strong and rank == 0:
        # bit-identity of the stitched table with the whole job on ONE GPU (A:896-915 is an order-free integer sum)
        one = Slot(whole)
        one.eng.set_cu_share(1)
        with torch.cuda.stream(one.stream):
            one.eng.reset()
            one.eng.process_device(one.rd, 0, one.dev_out)
            one.eng.sync()
        torch.cuda.synchronize()
        same = bool(torch.equal(one.table, table))
        assert same, "the all-reduced table of %d slices differs from the table of the whole job on one GPU" % world
        strong_check = "all-reduced table of %d slices == table of the whole %d-read job on one GPU (7 x %d uint32, bit-identical)" % (world, whole.n, G)
        one.eng.close()

    cpu = None
    if rank == 0 and world == 1 and args.cpu_passes >= 0:
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle
        hb = batch.to_host()
        # (a one-GPU box gets a 16-CPU share of its host, whatever the affinity mask says)
        cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("AMPLIPY_CPU_THREADS", "16")))
        cuts = [hb.n * k // cores for k in range(cores + 1)]
        got = table.cpu().numpy().view(np.uint32)
        h_pos = out_t["new_pos"].cpu().numpy()

        def shard(k, check=False):
            sb = hb.slice(cuts[k], cuts[k + 1])
            r = oracle.process(sb, G, mn, mx, mpl, 20, 4, read_base=cuts[k])
            if check:
                assert np.array_equal(r.trim.new_pos, h_pos[cuts[k]:cuts[k + 1]]), "trimmed positions differ from the oracle"
            return r.counts, r.events.size

        # all host cores: reads sharded over threads (the C call releases the GIL), private tables summed.
        # The first pass is the checker: the GPU result of the last step must equal the oracle's on the same batch.
        with ThreadPoolExecutor(cores) as pool:
            t1 = time.perf_counter()
            parts = list(pool.map(lambda k: shard(k, True), range(cores)))
            t1 = time.perf_counter() - t1
            total = np.sum([p[0] for p in parts], axis=0, dtype=np.uint32)
            assert np.array_equal(got[:G * 6].reshape(G, 6), total), "GPU count table differs from the oracle"
            assert int(got[G * 6:].sum()) == sum(p[1] for p in parts), "insertion events differ from the oracle"
            passes = args.cpu_passes if args.cpu_passes > 0 else max(1, min(40, int(10.0 / max(t1, 1e-3))))
            t_cpu = time.perf_counter()
            for _ in range(passes):
                parts = list(pool.map(shard, range(cores)))
            t_cpu = time.perf_counter() - t_cpu
        # one thread, on a sample
        ns = min(hb.n, 400000)
        t_one = time.perf_counter()
        oracle.process(hb.slice(0, ns), G, mn, mx, mpl, 20, 4)
        t_one = time.perf_counter() - t_one
        cpu = {"value": round(hb.n * passes / t_cpu, 1), "unit": "reads/s", "cores": cores, "kind": "port",
               "single_thread_value": round(ns / t_one, 1),
               "sample": "the full %d-read batch, %d passes of oracle/amplipy_oracle.c (trim + pileup) sharded by read over "
                         "%d threads with private count tables summed at the end (after one checking pass: GPU table, "
                         "trimmed positions and event count equal the oracle's); single_thread_value = %d reads on one thread"
                         % (hb.n, passes, cores, ns)}
        try:
            from oracle import py_restatement
            npy = min(hb.n, 100000)
            t_py = time.perf_counter()
            pr = py_restatement.process(hb.slice(0, npy), G, mn, mx, mpl, 20, 4)
            t_py = time.perf_counter() - t_py
            ref_py = oracle.process(hb.slice(0, npy), G, mn, mx, mpl, 20, 4)
            assert np.array_equal(pr, ref_py.counts), "pure-Python restatement differs from the C oracle"
            cpu["python_restatement"] = {"value": round(npy / t_py, 1), "unit": "reads/s", "cores": 1,
                                         "sample": "oracle/py_restatement.py (per-base interpreter loop, the shape of AmpliPy's own code) "
                                                   "on the first %d reads; counts equal the C oracle's" % npy}
        except ImportError:
            pass

    e2e = None
    if rank == 0 and world == 1 and not args.no_e2e:
        try:
            from tools import e2e_legs
            e2e = e2e_legs.measure(batch, genome, primers, ref_seq, dev)
        except Exception as ex:      # the legs are extras: never lose the bench line to them
            e2e = {"error": "%s: %s" % (type(ex).__name__, ex)}

    # ---- the other single-GPU configs of BASELINE.json, each as ONE launch checked against the oracle (outside the timed
    # region; the driver's one run then carries their roofline figures too) ---------------------------------------------
    extra = None
    if rank == 0 and world == 1 and not args.no_extra and not args.strong and args.depth == 10000:
        extra = {}
        try:
            from concurrent.futures import ThreadPoolExecutor
            from oracle import oracle
            cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("AMPLIPY_CPU_THREADS", "16")))

            def one_launch(db, hb, label):
                """db: DeviceBatch, hb: the same rows on the host -> dict with the scan pass's duration alone and its roofline
                fraction; asserts count table, trimmed positions and event count against the sharded oracle."""
                sl = Slot(db)
                sl.eng.set_cu_share(1)          # one launch alone: the whole chip
                ms = []
                for it in range(14):
                    with torch.cuda.stream(sl.stream):
                        sl.eng.reset()
                        sl.eng.process_device(sl.rd, 0, sl.dev_out)
                        sl.eng.sync()
                    ms.append(sl.eng.last_kernel_ms()[0])
                torch.cuda.synchronize()
                got = sl.table.cpu().numpy().view(np.uint32)
                h_pos = sl.out["new_pos"].cpu().numpy()
                h_st = sl.out["status"].cpu().numpy()
                n_out_ = int(sl.out["new_ncig"].sum().item())
                cuts = [hb.n * k // cores for k in range(cores + 1)]

                def shard(k):
                    r = oracle.process(hb.slice(cuts[k], cuts[k + 1]), G, mn, mx, mpl, 20, 4, read_base=cuts[k])
                    assert np.array_equal(r.trim.status, h_st[cuts[k]:cuts[k + 1]]), label + ": read statuses differ from the oracle"
                    okr = r.trim.status == 0
                    assert np.array_equal(r.trim.new_pos[okr], h_pos[cuts[k]:cuts[k + 1]][okr]), label + ": trimmed positions differ from the oracle"
                    return r.counts, r.events.size
                with ThreadPoolExecutor(cores) as pool:
                    parts = list(pool.map(shard, range(cores)))
                total = np.sum([p[0] for p in parts], axis=0, dtype=np.uint32)
                assert np.array_equal(got[:G * 6].reshape(G, 6), total), label + ": GPU count table differs from the oracle"
                assert int(got[G * 6:].sum()) == sum(p[1] for p in parts), label + ": insertion events differ from the oracle"
                lens = hb.lseq.astype(np.int64)
                n_cig = int(hb.cig_off[-1])
                ab = int(hb.n * (16 + 8) + ((lens + 1) // 2).sum() + lens.sum() + 4 * n_cig + 4 * n_out_ + 6 * G * 4 + 2 * G * 4)
                k_alone = float(np.mean(ms[-4:]))        # (the first ten launches of a fresh multi-GB batch run up to 15 % slower: see launch_ms)
                sl.eng.close()
                return {"reads": int(hb.n), "mean_read_len": round(float(lens.mean()), 1), "kernel_ms_alone": round(k_alone, 4),
                        "launch_ms": [round(x, 4) for x in ms], "ms_per_million_reads": round(k_alone / (hb.n / 1e6), 4), "algorithmic_bytes_per_launch": ab,
                        "achieved_GBs": round(ab / k_alone / 1e6, 1), "frac_alone": round(ab / k_alone / 1e6 / HBM_PEAK_GBS, 5),
                        "checked": "count table, read statuses, trimmed positions and event count equal oracle/amplipy_oracle.c on the same %d reads (sharded over %d threads)" % (hb.n, cores)}

            t_x = time.time()
            d3 = synth_torch.make_amplicon_batch_device(genome, amps, synth.reads_for_depth(100000), seed=1000, device=dev)
            extra["depth100k"] = one_launch(d3, d3.to_host(), "config 3 (100k x)")
            extra["depth100k"]["workload"] = "BASELINE config 3: the same genome at 100k x depth, 19,935,333 x 150 bp reads in ONE launch"
            del d3
            torch.cuda.empty_cache()
            h5 = synth.make_config5_batch(genome, amps, 200)
            d5 = synth_torch.DeviceBatch.from_host(h5, dev)
            extra["config5"] = one_launch(d5, h5, "config 5")
            extra["config5"]["workload"] = "BASELINE config 5: 8.0 M mixed 75-300 bp reads, long soft clips, indel-heavy CIGARs (50k x), ONE launch"
            del d5, h5
            torch.cuda.empty_cache()
            extra["seconds"] = round(time.time() - t_x, 1)
        except Exception as ex:      # the extras never cost the bench line
            extra["error"] = "%s: %s" % (type(ex).__name__, ex)

    if rank == 0:
        total_reads = n_reads * args.steps if not args.strong else job_reads * args.steps
        if not args.strong:
            total_reads *= world
        res = last["call"]
        line = {
            "metric": "aligned reads/s through trim+pileup+call, 30 kb ref @ %dk x depth" % (args.depth // 1000)
                      if args.depth % 1000 == 0 else "aligned reads/s through trim+pileup+call, 30 kb ref @ %d x depth" % args.depth,
            "value": round(total_reads / elapsed, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "u8/int32", "data": "synthetic",
            "config": {"batches": ("one batch per step in flight (%d distinct batches of the same generator, other seeds)" % (1 + len(slot_batches))) if not args.strong else "the rank's slice of the one job, shared by the steps in flight",
                       "workload": "synthetic 29,903 nt genome, 98-amplicon ARTIC-style primers, 150 bp paired reads, "
                                   + ("ONE job of %d x depth = %d reads cut into %d coordinate slices of equal base counts" % (args.depth, job_reads, world)
                                      if args.strong else "%d x depth = %d reads per GPU" % (args.depth, n_reads))
                                   + ", inputs resident in HBM",
                       "reads_per_gpu": n_reads, "read_len": L, "ref_len": G, "min_quality": 20, "window": 4,
                       "parallelism": "coordinate-range partition x%d + one RCCL all-reduce of the count table per step" % world,
                       "rccl_ranks": dist.get_world_size() if dist is not None else 0,
                       "steps_in_flight": depth, "work_streams": len(work_streams), "cu_share_of_a_pass": "1/%d" % cu_share, "untimed_steps_run": max(args.warmup, depth), "collective_on_own_stream": comm_stream is not None, "kernel_variant": args.variant, "error_reads": n_err,
                       "variants_called": res.n_records, "ins_relevant_positions": res.n_relevant,
                       "strong_check": strong_check},
            "roofline": {"bound": "hbm", "kernel": SCAN_KERNELS.get(args.variant, "?"),
                         "achieved": round(alg_bytes / (k_rate_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg_bytes / (k_rate_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "peak_measured": round(peak_measured, 1) if peak_measured else None,
                         "frac_of_measured": round(alg_bytes / (k_rate_ms * 1e-3) / 1e9 / peak_measured, 5) if peak_measured else None,
                         "frac_alone_of_measured": round(alg_bytes / k_solo / 1e6 / peak_measured, 5) if peak_measured else None,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(k_rate_ms, 5),
                         "kernel_ms_event_span": round(k_ms, 5),
                         "kernel_ms_alone": round(k_solo, 5),
                         "frac_alone": round(alg_bytes / k_solo / 1e6 / HBM_PEAK_GBS, 5),
                         "fast_kernel_ms_alone": round(float(np.mean(solo_fast)), 5),
                         "note": ("kernel_ms / achieved / frac: the timed region deals its steps over %d HIP streams, so the scan passes of "
                                  "consecutive steps overlap (the next one starts on the CUs the last blocks of this one have left) and "
                                  "the HIP-event span of one pass (kernel_ms_event_span) covers time it shares with its neighbours; the "
                                  "pass's rate inside the region is one pass per ms_per_step, which is what kernel_ms is here (an upper "
                                  "bound of its duration: the step's calling kernels are inside it). " % len(work_streams)
                                  if len(work_streams) > 1 else
                                  "kernel_ms / achieved / frac: HIP events around ALL kernels of the scan pass (amp_process_batch_device) "
                                  "inside the timed region, on the one work stream. ")
                                 + ("kernel_ms_alone / frac_alone: the pass with the GPU to itself and its grid sized for all CUs (5 passes after "
                                    "the timed region, HIP events on the engine's stream; profiles/ hold the rocprofv3 kernel trace of this "
                                    "leg); inside the region a pass is sized for 1/%d of the CUs (amp_set_cu_share) and passes run side by side; "
                                    "fast_kernel_ms_alone: the first kernel of the pass alone (k_fast for variant 4). " % cu_share)
                                 + "peak_measured: a 1 GiB device-to-device copy on this box (read + written bytes per second), frac_of_measured / "
                                   "frac_alone_of_measured: the same two fractions against it.  traffic: HBM bytes per launch from the rocprofv3 "
                                   "counter passes committed under profiles/ (a constant of the profile, not measured in this run: counters need the profiler)"},
            "cpu_baseline": cpu,
            "e2e": e2e,
            "extra": extra,
            "gen_seconds": round(t_gen, 2),
        }
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    for sl in slots:
        sl.eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
