"""Minimal driver for profiling the scan kernels (development aid): builds the bench workload in HBM and runs
amp_process_batch_device a few times.  usage: run_scan.py [--depth D] [--iters K] [--variant V] [--check]"""
import argparse, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amplipy_amd import abi, lib, synth, synth_torch
ap = argparse.ArgumentParser()
ap.add_argument("--depth", type=int, default=10000); ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--variant", type=int, default=4); ap.add_argument("--check", action="store_true")
ap.add_argument("--indel-frac", type=float, default=0.10)
ap.add_argument("--read-len", type=int, default=150)
ap.add_argument("--window", type=int, default=4); ap.add_argument("--min-quality", type=int, default=20)
ap.add_argument("--same-rows", action="store_true", help="every read points at the bytes of read 0 (timing experiment: no HBM traffic for bases / qualities; results are wrong on purpose)")
a = ap.parse_args()
g = synth.make_genome(); primers, amps = synth.make_artic_scheme(); G = g.size
n = synth.reads_for_depth(a.depth, a.read_len)
b = synth_torch.make_amplicon_batch_device(g, amps, n, 1000, "cuda:0", read_len=a.read_len, indel_frac=a.indel_frac); torch.cuda.synchronize()
if a.same_rows:
    b.seq_off8.zero_()
mn, mx, mpl = lib.find_overlapping_primers(G, [(s, e) for s, e, _ in primers], 0)
e = lib.Engine(G); e.set_kernel_variant(a.variant); e.set_timing(bool(os.environ.get("AMP_SPLIT")))
e.set_primers(mn, mx, mpl); e.set_params(a.min_quality, a.window, True, True); e.reserve_events(max(1 << 20, n // 4))
out = {k: torch.zeros(sz, dtype=dt, device="cuda:0") for k, sz, dt in
       (("new_pos", n, torch.int32), ("new_ncig", n, torch.int32), ("new_cig", b.n_cig + 3 * n, torch.int32),
        ("ref_len", n, torch.int32), ("trim_flags", n, torch.uint8), ("status", n, torch.uint8))}
dev_out = abi.AmpTrimOut(*[out[k].data_ptr() for k in ("new_pos", "new_ncig", "new_cig", "ref_len", "trim_flags", "status")])
rd = b.struct()
ms = []
for it in range(a.iters):
    e.reset(); e.process_device(rd, 0, dev_out); e.sync(); ms.append(e.last_kernel_ms())
print("variant %d window %d depth %d reads %d: total/scan ms per launch:" % (a.variant, a.window, a.depth, n), ["%.3f/%.3f" % m for m in ms])
if not os.environ.get("AMP_STAMPS"): print("general-pass reads of the last launch:", int(e.debug_counters()[7]))
if os.environ.get("AMP_F6_WAITSTAMPS"):
    dc = e.debug_counters(); turns = max(int(dc[6]), 1)
    print("k_fast6 waits (last launch): turns %d; cycles per turn: wait in front of pass 1b %.0f, wait at the end of the turn %.0f, drain of the adds in front of it %.0f" % (turns, int(dc[8]) / turns, int(dc[9]) / turns, int(dc[10]) / turns))
if os.environ.get("AMP_F6_STAMPS"):
    dc = e.debug_counters(); turns = max(int(dc[6]), 1)
    names = ["requests+anchor", "pass 1a", "wait", "clips+results", "pass 1b", "requests+range ends", "pass 2", "tail+end wait"]
    tot = float(sum(int(dc[8 + k]) for k in range(8)))
    print("k_fast6 stamps (last launch): turns %d; shader cycles per turn: %s | total %.0f" % (turns, ", ".join("%s %.0f" % (names[k], int(dc[8 + k]) / turns) for k in range(8)), tot / turns))
if os.environ.get("AMP_STAMPS") not in (None, "", "0") and not os.environ.get("AMPLIHIP_PHASES"):
    dc = e.debug_counters()
    tot = float(sum(int(x) for x in dc[9:15])) or 1.0
    nt = (n + 63) // 64
    nw = 8 * ((n + 8191) // 8192)
    print("k_fast whole kernel per wave: %.0f shader cycles = %.1f us of the 100 MHz clock -> shader clock %.0f MHz; tile loop %.0f cycles of them"
          % (int(dc[8]) / nw, int(dc[15]) / nw / 100.0, 100.0 * int(dc[8]) / max(int(dc[15]), 1), tot / nw))
    M = (1 << 64) - 1
    print("k_fast waves: shortest %.1f us, longest %.1f us; first start to last end %.1f us" % ((M - int(dc[5])) / 100.0, int(dc[4]) / 100.0, (int(dc[7]) - (M - int(dc[6]))) / 100.0))
    w = e.debug_blocks().reshape(-1, 8)[:nw // 8].astype(np.int64)
    dur = w[:, 0].reshape(-1, 1) / 100.0; st = (w[:, 1] - w[:, 1].min()).reshape(-1, 1) / 100.0
    bd = dur.max(axis=1)
    print("per block (us): start offset min/max %.1f/%.1f; duration min %.1f median %.1f max %.1f; waves of one block differ by up to %.1f"
          % (st.min(), st.max(), bd.min(), np.median(bd), bd.max(), (dur.max(axis=1) - dur.min(axis=1)).max()))
    print("block duration by block index modulo 8 (XCD):", [round(float(bd[k::8].mean()), 1) for k in range(8)])
    print("block durations in block order:", [int(x) for x in bd])
    rpb = 8192
    pos_h = b.pos.cpu().numpy().astype(np.int64); fl = b.flag.cpu().numpy().astype(np.int64) & 0xFFFF
    nb_ = bd.size
    feat = {"reverse fraction": [], "position span": [], "distinct starts": [], "first position": []}
    for k in range(nb_):
        pp = pos_h[k * rpb:(k + 1) * rpb]; ff = fl[k * rpb:(k + 1) * rpb]
        feat["reverse fraction"].append(float(((ff & 16) != 0).mean())); feat["position span"].append(float(pp[-1] - pp[0]))
        feat["distinct starts"].append(float(np.unique(pp).size)); feat["first position"].append(float(pp[0]))
    for k_, v in feat.items():
        print("  corr(duration, %s) = %.2f" % (k_, np.corrcoef(bd, np.array(v))[0, 1]))
    ph = w[:, 2:8].astype(np.float64)                                    # wave 0 of every block: cycles per tile (16 tiles, stored >> 4)
    so = np.argsort(bd)
    print("  phases (cycles per tile: top wait, rows+clips+issue, scan, qclip, count, careful) of the 30 fastest blocks:", [int(x) for x in ph[so[1:31]].mean(axis=0)])
    print("  ... of the 30 slowest blocks:", [int(x) for x in ph[so[-30:]].mean(axis=0)])
    raw = e.debug_blocks().reshape(-1)
    pw = raw[2048:2048 + nw * 6].reshape(-1, 8, 6).astype(np.float64)
    for bidx in list(so[-4:]) + list(so[1:3]):
        print("  block %d (%.0f us): per wave: stamped loop us %s | older stores done at %s | then the phantom tile's quality DMA %s | its base loads %s | last stores %s | last fold %s" % ((bidx, bd[bidx], [int(x * 16 / 2.29e3) for x in pw[bidx, :, 0]]) + tuple([int(x / 100) for x in pw[bidx, :, k]] for k in range(1, 6))))
    slow = np.argsort(bd)[-6:]; fast = np.argsort(bd)[:6]
    for name, idx in (("slowest", slow), ("fastest", fast)):
        print("  %s blocks:" % name, [(int(i), int(bd[i]), round(feat["reverse fraction"][i], 2), int(feat["position span"][i]), int(feat["distinct starts"][i])) for i in idx])
    print("k_fast phases, shader cycles per tile and wave (last launch): top wait %.0f | rows, primer clips, next tile issued %.0f | scan %.0f | quality clip, results %.0f | count %.0f | careful, hand-over %.0f | total %.0f"
          % tuple([int(dc[8 + k]) / nt for k in (1, 2, 3, 4, 5, 6)] + [tot / nt]))
if os.environ.get("AMPLIHIP_PHASES"):
    dc = e.debug_counters()
    tot = float(sum(int(x) for x in dc[8:13])) or 1.0
    print("general pass (k_tile<LIST>) stamps: tiles %d; cycles per tile P1 %.0f P2 %.0f P3 %.0f P4 %.0f tail %.0f; waves %d mean loop %.0f cyc, wait at final barrier %.0f cyc; block mean %.0f cyc"
          % (int(dc[13]), *[int(dc[8 + k]) / max(int(dc[13]), 1) for k in range(5)], int(dc[15]), int(dc[6]) / max(int(dc[15]), 1), int(dc[7]) / max(int(dc[15]), 1), int(dc[4]) / max(int(dc[15]) // 8, 1)))
if a.check:
    from oracle import oracle
    hb = b.to_host()
    ref = oracle.process(hb, G, mn, mx, mpl, a.min_quality, a.window)
    assert np.array_equal(e.counts(), ref.counts), "counts differ"
    assert np.array_equal(out["new_pos"].cpu().numpy(), ref.trim.new_pos)
    assert np.array_equal(out["new_ncig"].cpu().numpy().view(np.uint32), ref.trim.new_ncig)
    order = ["ref_pos", "read", "q_from", "q_to"]
    assert np.array_equal(np.sort(e.events(), order=order), np.sort(ref.events, order=order)), "events differ"
    print("check ok")
