#!/usr/bin/env python3
"""Randomised campaign for the device's pair walk (tredgpu_inflate_walk) against the host's file layer (GPU box).

Every round simulates a few samples (random coverage, loci subset, seed), perturbs the records' flags (duplicates,
unpaired reads, flipped strands, unmapped reads), writes each as a BAM whose BGZF blocks are cut at a random size WITHOUT
regard to record boundaries (from 200 bytes -- every record straddles blocks -- to 64 KiB) and compares, per locus: the two
pair-length lists with tredbam_pe_lengths, and the whole scan with the walks' results handed in -- pair lengths, window
offsets, and the alternative loci's records (tredbam_scan_walked over the fetched blocks only) -- with the plain scan.  Prints one JSON line.

usage: python tools/fuzz_walk.py [rounds = 20] [seed = 1]
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


# optional fields behind every record of every third round (as aligners leave them); the array's bytes read as a record's head
AUX = b"NMC\x02MDZ75A74\x00ASC\x91RGZgroup1\x00XSC\x13" + b"ZBBC" + (40).to_bytes(4, "little") + bytes([40, 0, 0, 0, 3, 0, 0, 0, 5, 0, 0, 0, 2, 0, 73, 18] + [0] * 24)


def main():
    from tredparse_amd import _lib, bamio, synth, synth_bam
    from tredparse_amd.bam_parser import DNAPE_ELONGATE, FLANKMATCH, SPAN, _site_arrays, walk_need
    from tredparse_amd.meta import TREDsRepo
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    repo = TREDsRepo()
    all_loci = synth.load_loci()
    root = tempfile.mkdtemp(prefix="tred_fuzzwalk_")
    inf = _lib.Inflater(0)
    out = {"rounds": rounds, "samples": 0, "regions": 0, "regions_walked": 0, "declined": {}, "pairs": 0, "window_records": 0, "blocks": 0,
           "blocks_fetched": 0, "mismatching_regions": 0, "mismatching_scans": 0, "alt_regions": 0, "alt_regions_declined": 0, "alt_records": 0, "scans_that_inflated_blocks": 0, "regions_chained_serially": 0, "block_sizes": []}
    t0 = time.time()
    for rnd in range(rounds):
        loci = [all_loci[i] for i in sorted(rng.choice(len(all_loci), size=int(rng.integers(2, 7)), replace=False))]
        names = [l["name"] for l in loci]
        cases = []
        for k in range(int(rng.integers(1, 4))):
            cov = float(rng.choice([3, 10, 30, 60]))
            recs, _ = synth_bam.simulate_sample(int(rng.integers(1 << 30)), loci, synth.SynthParams(coverage=cov, expanded_max=120, expanded_frac=0.3))
            n = len(recs.flag)
            recs.flag[rng.random(n) < 0.03] |= 0x400                       # duplicates
            recs.flag[rng.random(n) < 0.02] &= ~0x1                        # unpaired
            recs.flag[rng.random(n) < 0.05] ^= 0x10                        # strand flipped
            block = int(rng.choice([200, 333, 1000, 4096, 20000, 0xff00]))
            path = os.path.join(root, "r{}_{}.bam".format(rnd, k))
            # every fifth round: base qualities that look like record heads (the lanes' guesses go wrong, the serial chain takes over)
            synth_bam.write_bam(path, recs, sample="f{}_{}".format(rnd, k), block=block, split_records=True,
                                decoys=0.5 if rnd % 5 == 4 else 0.0, decoy_seed=rnd, aux=AUX if rnd % 3 == 2 else b"")
            cases.append(path)
            out["block_sizes"].append(block)
        handles = [bamio.AlignmentFile(p) for p in cases]
        sites_of, regions_of, plans = [], [], []
        for f in handles:
            sites, regions = _site_arrays(repo, names, [repo[n] for n in names], f)
            plans.append(f.plan(sites, regions, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN))
            sites_of.append(sites); regions_of.append(regions)
        n_all = sum(p[0] for p in plans)
        comp, _, coff, ooff = inf.reserve(sum(p[1] for p in plans), sum(p[2] for p in plans), n_all)
        at = cb = ob = c0 = 0
        firsts, tasks, chunks, blk, atasks, achunks, a0 = [], [], [], [], [], [], 0
        for f, sites, regions, p in zip(handles, sites_of, regions_of, plans):
            f.plan_fill(inf.comp_addr, cb, ob, coff[at:at + p[0] + 1], ooff[at:at + p[0] + 1])
            t, c = f.plan_walks(sites, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
            t, c = t.copy(), c.copy()
            t["chunk_first"] += c0; t["block_first"] += at; t["block_end"] += at
            c["begin_block"][c["begin_block"] >= 0] += at
            ta, ca = f.plan_alt_walks(sites, regions, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
            ta, ca = ta.copy(), ca.copy()
            ta["chunk_first"] += a0; ta["block_first"] += at; ta["block_end"] += at
            ca["begin_block"][ca["begin_block"] >= 0] += at
            atasks.append(ta); achunks.append(ca); a0 += len(ca)
            tasks.append(t); chunks.append(c); blk.append(f.plan_blocks()); firsts.append(at)
            at, cb, ob, c0 = at + p[0], cb + p[1], ob + p[2], c0 + len(c)
        bcoff, bclen, bcrc = (np.concatenate([b[k] for b in blk]) for k in range(3))
        status, crc, res, gp, tp, ares, alt_need = inf.run_walk(n_all, bcoff, bclen, bcrc, np.concatenate(tasks), np.concatenate(chunks), pairs_per_task=16384,
                                                                alt_tasks=np.concatenate(atasks), alt_chunks=np.concatenate(achunks))
        walkable = np.concatenate(atasks)["n_chunks"] >= 0
        out["alt_regions"] += int(walkable.sum())
        out["alt_regions_declined"] += int((ares["status"][walkable] != 0).sum())
        out["regions_chained_serially"] += inf.walk_serial_regions()
        out["alt_records"] += int(ares["n"][ares["status"] == 0].sum())
        assert (status == 0).all() and (crc == bcrc).all()
        out["blocks"] += n_all
        t_at = a_at = 0
        for f, path, sites, regions, p, first, b, ta in zip(handles, cases, sites_of, regions_of, plans, firsts, blk, atasks):
            r = res[t_at:t_at + len(names)]
            t_at += len(names)
            ar = ares[a_at:a_at + len(ta)]
            a_at += len(ta)
            plain = bamio.AlignmentFile(path)
            units, pools = plain.scan(sites, regions, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
            for k, name in enumerate(names):
                t = repo[name]
                out["regions"] += 1
                if r["status"][k] != 0:
                    out["declined"][str(int(r["status"][k]))] = out["declined"].get(str(int(r["status"][k])), 0) + 1
                    continue
                out["regions_walked"] += 1
                eg, et = plain.pe_lengths(t.chr, t.repeat_start - DNAPE_ELONGATE, t.repeat_end + DNAPE_ELONGATE,
                                          t.repeat_start - FLANKMATCH, t.repeat_end + FLANKMATCH, SPAN)
                g = gp[r["global_first"][k]:r["global_first"][k] + r["n_global"][k]]
                tt = tp[r["target_first"][k]:r["target_first"][k] + r["n_target"][k]]
                out["pairs"] += len(eg) + len(et)
                out["window_records"] += int(r["n_window"][k])
                if list(g) != eg or list(tt) != et:
                    out["mismatching_regions"] += 1
            need = walk_need(b[0], b[3], r, alt_need[first:first + p[0]])
            full = np.zeros(n_all, np.uint8)
            full[first:first + p[0]] = need
            inf.fetch(full)
            out["blocks_fetched"] += int(need.sum())
            f.preload(inf.out_addr, ooff[first:first + p[0] + 1], np.where(need != 0, status[first:first + p[0]], 1).astype(np.int32), crc[first:first + p[0]])
            u2, p2 = f.scan(sites, regions, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN, pe=(r, gp, tp), alt=ar)
            hits, misses = f.preload_clear()
            same = all((units[key] == u2[key]).all() for key in units.dtype.names) and all(
                (pools[key] == p2[key]) if isinstance(pools[key], bytes) else np.array_equal(pools[key], p2[key]) for key in pools)
            out["mismatching_scans"] += 0 if same else 1
            out["scans_that_inflated_blocks"] += 1 if (misses and (r["status"] == 0).all() and (ar["status"][ta["n_chunks"] >= 0] == 0).all()) else 0
            out["samples"] += 1
            plain.close(); f.close()
        for path in cases:
            os.remove(path); os.remove(path + ".bai")
    out["seconds"] = round(time.time() - t0, 1)
    out["block_sizes"] = sorted(set(out["block_sizes"]))
    out["library"] = _lib.version()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
