#!/usr/bin/env python3
"""TRED caller CLI -- drop-in for tredparse/tred.py (same flags, same JSON keys / formatting, same VCF
lines), with the per-read Smith-Waterman and the (h1,h2) likelihood grid running on an MI355X through
libtredgpu.so.  All loci of a sample (and, with --batch-samples, of several samples) are genotyped
in one GPU batch instead of one locus at a time.

Mirrors: set_argparse tred.py:64-113, run :180-278, vcfstanza :281-293, to_json :296-313,
to_vcf :316-374, read_csv :401-440, main :451-539.  Not carried over: S3 push (--output_path) and the
HLI-internal "@sample" lookup (:377-398) -- both outside the hot path.
"""
import argparse
import gzip
import json
import logging
import os
import os.path as op
import shutil
import sys
import time
from datetime import datetime as dt, timedelta

from . import __version__
from .bam_parser import BamDepth, BamParser, BamParserResults, BamReadLen, SPAN, read_alignment
from .meta import TREDsRepo
from .models import IntegratedCaller
from .utils import InputParams, mkdir

logging.basicConfig()
logger = logging.getLogger(__name__)

INFO = """##INFO=<ID=RPA,Number=1,Type=String,Description="Repeats per allele">
##INFO=<ID=END,Number=1,Type=Integer,Description="End position of variant">
##INFO=<ID=MOTIF,Number=1,Type=String,Description="Canonical repeat motif">
##INFO=<ID=NS,Number=1,Type=Integer,Description="Number of samples with data">
##INFO=<ID=REF,Number=1,Type=Integer,Description="Reference copy number">
##INFO=<ID=CR,Number=1,Type=Integer,Description="Disease copy number cutoff">
##INFO=<ID=IH,Number=1,Type=String,Description="Inheritance">
##INFO=<ID=RL,Number=1,Type=Integer,Description="Reference STR track length in bp">
##INFO=<ID=VT,Number=1,Type=String,Description="Variant type">
##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">
##FORMAT=<ID=GA,Number=1,Type=String,Description="Genotype with absolute copy numbers">
##FORMAT=<ID=FR,Number=1,Type=String,Description="Full spanning reads aligned to locus">
##FORMAT=<ID=PR,Number=1,Type=String,Description="Partial reads aligned to locus">
##FORMAT=<ID=RR,Number=1,Type=String,Description="Repeat-only reads aligned to locus">
##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Mean read depth around locus">
##FORMAT=<ID=FDP,Number=1,Type=Integer,Description="Full spanning read depth">
##FORMAT=<ID=PDP,Number=1,Type=Integer,Description="Partial read depth">
##FORMAT=<ID=RDP,Number=1,Type=Integer,Description="Repeat read depth">
##FORMAT=<ID=PEDP,Number=1,Type=Integer,Description="Paired-end read depth">
##FORMAT=<ID=CI,Number=1,Type=String,Description="95% conf interval of estimates">
##FORMAT=<ID=PP,Number=1,Type=Float,Description="Posterior probability of disease">
##FORMAT=<ID=LABEL,Number=1,Type=String,Description="Risk assessment">
"""


class DefaultHelpParser(argparse.ArgumentParser):
    def error(self, message):
        sys.stderr.write('error: {}\n\n'.format(message))
        sys.exit(not self.print_help())


def set_argparse():
    TRED_NAMES = TREDsRepo().names
    p = DefaultHelpParser(description=__doc__, prog="tred.py",
                          formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('infile', nargs='?', help="Input path (BAM, list of BAMs, or csv format)")
    p.add_argument('--ref', help='Reference genome version',
                   choices=("hg38", "hg38_nochr", "hg19", "hg19_nochr"), default='hg38')
    p.add_argument('--tred', help='STR disorder, default is to run all', action='append',
                   choices=sorted(TRED_NAMES), default=None)
    p.add_argument('--haploid', help='Treat these chromosomes as haploid', action='append')
    p.add_argument('--useclippedreads', default=False, action="store_true", help='Include clipped reads in inference')
    p.add_argument('--noalts', default=False, action="store_true",
                   help='Do not scan extra sites for mismapped reads, faster but less accurate')
    p.add_argument('--norepeatpairs', default=False, action="store_true",
                   help='Exclude pairs of repeat-only reads from evidence')
    p.add_argument('--log', choices=("INFO", "DEBUG"), default="INFO", help='Print debug logs, DEBUG=verbose')
    p.add_argument('--version', action='version', version="%(prog)s " + __version__)
    p.add_argument('--toy', help=argparse.SUPPRESS, action="store_true")
    g = p.add_argument_group("Performance options")
    g.add_argument('--cpus', help='Host workers for BAM reading (the GPU batch is shared)', type=int, default=1)
    g.add_argument('--gpu', help='GPU index', type=int, default=0)
    g.add_argument('--maxinsert', default=300, type=int, help="Maximum number of repeats")
    g.add_argument('--fullsearch', default=False, action="store_true", help="Full grid search, could be slow")
    g = p.add_argument_group("I/O options")
    g.add_argument("--workdir", default=os.getcwd(), help="Specify work dir")
    g.add_argument('--cleanup', default=False, action="store_true", help="Cleanup the workdir after done")
    g.add_argument('--checkexists', default=False, action="store_true", help="Do not run if JSON output exists")
    g.add_argument('--no-output', default=False, action="store_true", help="Do not write JSON and VCF output")
    g = p.add_argument_group("AWS and Docker options")
    g.add_argument("--sample_id", help="Sample ID")
    g.add_argument("--workflow_execution_id", help="Workflow execution ID")
    g.add_argument("--input_bam_path", help="Input path, override infile")
    g.add_argument("--output_path", help="(S3 push of the reference is not carried over; ignored)")
    return p


def bam_path(bam):
    if bam.startswith(("s3://", "http://", "ftp://", "https://")):
        return bam
    return op.abspath(bam)


def check_bam(bam):
    try:
        read_alignment(bam).close()
    except (IOError, ValueError) as e:
        logger.error("Cannot retrieve file `{}` ({})".format(bam, e))
        return None
    return bam


def counter_s(c):
    return ";".join(["{}|{}".format(k, int(v)) for k, v in sorted(c.items())])


class _Pending(object):
    """One sample x locus unit between host collection and the GPU batch."""
    __slots__ = ("tred", "bp", "caller", "depth", "ip")


def collect_sample(arg):
    """Host half of run() (tred.py:180-249): open the BAM, infer sex / read length / depth, select reads
    and pair lengths for every locus.  Returns (result skeleton, [pending units])."""
    samplekey, bam, repo, tredNames, maxinsert, fullsearch, clip, alts, repeatpairs, log = arg
    gender, ydepth = 'Unknown', -1
    tredCalls = {"inferredGender": gender, "depthY": ydepth}
    result = {'samplekey': samplekey, 'bam': bam, 'tredCalls': tredCalls}
    if check_bam(bam) is None:
        return result, []
    if any(repo[tred].is_xlinked for tred in tredNames):   # infer gender from depth on chrY (:201-213)
        try:
            ydepth = BamDepth(bam, repo.ref, logger).get_Y_depth()
            gender = 'Male' if ydepth > 1 else 'Female'
        except Exception:
            pass
        tredCalls["inferredGender"] = gender
        tredCalls["depthY"] = float(ydepth) if ydepth != -1 else ydepth
    READLEN = 150
    try:
        READLEN = BamReadLen(bam, logger).readlen
    except Exception:
        pass
    tredCalls["readLen"] = READLEN
    pending = []
    for tred in tredNames:
        bd = BamDepth(bam, repo.ref, logger)
        xtred = repo[tred]
        WINDOW_START = max(0, xtred.repeat_start - SPAN)
        WINDOW_END = xtred.repeat_end + SPAN
        try:
            depth = bd.region_depth(xtred.chr, WINDOW_START, WINDOW_END)
        except Exception as e:
            depth = 30
            logger.error("Exception on `{}` {} ({}). Set depth={}".format(bam, tred, e, depth))
        ip = InputParams(bam=bam, READLEN=READLEN, tredName=tred, repo=repo, maxinsert=maxinsert,
                         fullsearch=fullsearch, gender=gender, depth=depth, clip=clip, alts=alts,
                         repeatpairs=repeatpairs, log=log)
        try:
            bp = BamParser(ip)
            bp.collect()
            caller = IntegratedCaller(bp, maxinsert=maxinsert, fullsearch=fullsearch)   # runs PEextractor
        except Exception as e:   # the reference drops the locus on any error (:245-249)
            logger.error("Exception on `{}` {} ({})".format(bam, tred, e))
            continue
        u = _Pending()
        u.tred, u.bp, u.caller, u.depth, u.ip = tred, bp, caller, depth, ip
        pending.append(u)
    return result, pending


def finish_sample(result, pending, unit_results):
    """Second half of run() (tred.py:251-275): fill tredCalls from the GPU results."""
    tredCalls = result['tredCalls']
    for u, res in zip(pending, unit_results):
        tred, bp, caller = u.tred, u.bp, u.caller
        try:
            bp.finish(res.tags, res.hs)
            caller.counts, caller.rept = bp.counts, bp.rept
            caller.from_result(res)
        except Exception as e:
            logger.error("Exception on `{}` {} ({})".format(result['bam'], tred, e))
            continue
        tpResult = BamParserResults(u.ip, bp, caller)
        alleles = tpResult.alleles
        tredCalls[tred + ".1"] = alleles[0]  # .1 is the shorter allele
        tredCalls[tred + ".2"] = alleles[1]  # .2 is the longer allele
        tredCalls[tred + ".FR"] = counter_s(tpResult.counts["FULL"])
        tredCalls[tred + ".PR"] = counter_s(tpResult.counts["PREF"])
        tredCalls[tred + ".RR"] = counter_s(tpResult.counts["REPT"])
        tredCalls[tred + ".DP"] = u.depth
        tredCalls[tred + ".FDP"] = tpResult.FDP
        tredCalls[tred + ".PDP"] = tpResult.PDP
        tredCalls[tred + ".RDP"] = tpResult.RDP
        tredCalls[tred + ".PEDP"] = tpResult.PEDP
        tredCalls[tred + ".PEG"] = tpResult.PEG
        tredCalls[tred + ".PET"] = tpResult.PET
        tredCalls[tred + ".CI"] = tpResult.CI
        tredCalls[tred + ".PP"] = tpResult.PP
        tredCalls[tred + ".label"] = tpResult.label
        tredCalls[tred + ".details"] = tpResult.details
        tredCalls[tred + ".P_h1"] = tpResult.P_h1
        tredCalls[tred + ".P_h2"] = tpResult.P_h2
        tredCalls[tred + ".P_h1h2"] = tpResult.P_h1h2
        tredCalls[tred + ".P_PEG"] = tpResult.P_PEG
        tredCalls[tred + ".P_PET"] = tpResult.P_PET
    return result


def units_of(arg, pending):
    """The engine.Unit of every pending sample x locus of one sample (arg = the run() argument tuple)."""
    clip, repeatpairs = arg[6], arg[8]
    units = []
    for u in pending:
        unit = u.caller.unit([s for _, s in u.bp.reads])
        if not (repeatpairs or clip):   # --norepeatpairs: mates share a query name (bam_parser.py:270-287)
            ids = {}                    # (the device removes REPT/REPT pairs from the histograms; finish() on details)
            unit.read_pair_ids = [ids.setdefault(name, len(ids)) for name, _ in u.bp.reads]
        units.append(unit)
    return units


def run(arg, engine=None):
    """Run the TRED caller on one sample (same argument tuple and return value as tred.py:180-278)."""
    from .engine import Engine
    engine = engine or Engine()
    result, pending = collect_sample(arg)
    units = units_of(arg, pending)
    res = engine.genotype(units) if units else []
    return finish_sample(result, pending, res)


def host_pool(cpus, n_tasks):
    """Worker processes for the host half (the reference's Pool over samples, tred.py:521-532).  Create it BEFORE
    the Engine: the workers are forked and must not inherit an initialised GPU runtime."""
    if cpus > 1 and n_tasks > 1:
        import multiprocessing
        return multiprocessing.get_context("fork").Pool(min(cpus, n_tasks))
    return None


def run_many(task_args, engine, pool=None, batch=64, sink=None):
    """run() over many samples: the host half of up to `batch` samples is collected (by the pool's workers, which
    never touch the GPU), their units go to the GPU as ONE batch, and each finished result is handed to
    sink(result) (or returned as a list).  The pool stays the caller's (close and join it when done)."""
    out = []
    chunks = [task_args[i:i + batch] for i in range(0, len(task_args), batch)]
    ahead = pool.map_async(collect_sample, chunks[0]) if pool and chunks else None
    for k, chunk in enumerate(chunks):
        if pool:
            collected = ahead.get()
            # the workers read the next batch's BAMs while this one is on the GPU and being formatted
            ahead = pool.map_async(collect_sample, chunks[k + 1]) if k + 1 < len(chunks) else None
        else:
            collected = [collect_sample(a) for a in chunk]
        units, spans = [], []
        for a, (_, pending) in zip(chunk, collected):
            us = units_of(a, pending)
            spans.append((len(units), len(units) + len(us)))
            units += us
        res = engine.genotype(units) if units else []
        for (result, pending), (lo, hi) in zip(collected, spans):
            r = finish_sample(result, pending, res[lo:hi])
            if sink is not None:
                sink(r)
            else:
                out.append(r)
    return out


def vcfstanza(sampleid, bam, tredCalls, ref):
    m = "##fileformat=VCFv4.1\n"
    m += "##fileDate={}{:02d}{:02d}\n".format(dt.now().year, dt.now().month, dt.now().day)
    m += "##source={} {}\n".format(__file__, bam)
    m += "##reference={}\n".format(ref)
    m += "##inferredGender={} depthY={}\n".format(tredCalls["inferredGender"], tredCalls["depthY"])
    m += "##readLen={}bp\n".format(tredCalls["readLen"])
    m += INFO
    header = "CHROM POS ID REF ALT QUAL FILTER INFO FORMAT\n".split() + [sampleid]
    m += "#" + "\t".join(header)
    return m


def to_json(results, ref, repo, treds=("HD",), store=None, quiet=False):
    sampleid = results['samplekey']
    calls = results['tredCalls']
    if not calls:
        return
    jsonfile = ".".join((sampleid, "json"))
    js = json.dumps(results, sort_keys=True, indent=4, separators=(',', ': '))
    if not quiet:
        print(js)
    with open(jsonfile, "w") as fw:
        print(js, file=fw)


def to_vcf(results, ref, repo, treds=("HD",), store=None):
    registry = {tred: repo.get_info(tred) for tred in treds}
    sampleid, bam, calls = results['samplekey'], results['bam'], results['tredCalls']
    if not calls:
        return
    vcffile = ".".join((sampleid, "tred.vcf.gz"))
    contents = []
    for tred in treds:
        if tred + ".1" not in calls:
            continue
        a, b = calls[tred + ".1"], calls[tred + ".2"]
        chr, start, ref_copy, repeat, info = registry[tred]
        alleles = set([a, b])
        refv = set([ref_copy])
        rpa = sorted(alleles - refv)
        alt = ",".join(x * repeat for x in rpa) if (rpa and rpa[0] != -1) else "."
        if rpa:
            info += ";RPA={}".format(",".join((str(x) for x in rpa)))
            if ref_copy in alleles:
                gt = "0/1"
            elif len(rpa) == 1:
                gt = "1/1"
            else:
                gt = "1/2"
        else:
            gt = "0/0"
        gb = "{}/{}".format(a, b)
        fields = "{}:{}:{}:{}:{}:{}:{}:{}:{}:{}:{}:{:.4g}:{}".format(
            gt, gb, calls[tred + ".FR"], calls[tred + ".PR"], calls[tred + ".RR"], calls[tred + ".DP"],
            calls[tred + ".FDP"], calls[tred + ".PDP"], calls[tred + ".RDP"], calls[tred + ".PEDP"],
            calls[tred + ".CI"], calls[tred + ".PP"], calls[tred + ".label"])
        m = "\t".join(str(x) for x in (chr, start, tred, ref_copy * repeat, alt, ".", ".", info,
                                       "GT:GB:FR:PR:RR:DP:FDP:PDP:RDP:PEDP:CI:PP:LABEL", fields))
        contents.append((chr, start, m))
    with gzip.open(vcffile, "wt") as fw:
        print(vcfstanza(sampleid, bam, calls, ref), file=fw)
        contents.sort()
        for chr, start, m in contents:
            print(m, file=fw)


def read_csv(csvfile, args):
    if csvfile[0] == '@':
        raise SystemExit("the HLI-internal @sample lookup of the reference (tred.py:377-398) is not available")
    if csvfile.endswith(".bam") or csvfile.endswith(".cram"):   # Mode 1: a single BAM
        bam = bam_path(csvfile)
        if args.workflow_execution_id and args.sample_id:
            samplekey = "_".join((args.workflow_execution_id, args.sample_id))
        else:
            samplekey = op.basename(bam).rsplit(".", 1)[0]
        return [(samplekey, bam, None)]
    with open(csvfile) as fp:
        lines = fp.read().splitlines()
    contents = []
    header = lines[0].strip() if lines else ""
    if header.endswith(".bam") and header.count(",") == 0:      # Mode 2: list of BAM files
        for row in lines:
            bam = bam_path(row.strip())
            contents.append((op.basename(bam).rsplit(".", 1)[0], bam, None))
        return contents
    for row in lines:                                           # Mode 3: CSV
        atoms = row.strip().split(",")
        if len(atoms) < 2:
            continue
        samplekey, bam = atoms[:2]
        tred = atoms[2] if len(atoms) == 3 else None
        bam = bam_path(bam)
        if bam.endswith(".bam"):
            contents.append((samplekey, bam, tred))
    return contents


def write_vcf_json(results, ref, repo, treds, store, quiet=False):
    try:
        to_vcf(results, ref, repo, treds=treds, store=store)
        to_json(results, ref, repo, treds=treds, store=store, quiet=quiet)
    except Exception as e:
        print("Error writing: {} ({})".format(results, e), file=sys.stderr)


def main(args, quiet=False):
    p = set_argparse()
    args = p.parse_args(args)
    logger.setLevel(getattr(logging, args.log.upper(), "INFO"))
    start = time.time()
    workdir = args.workdir
    cwd = os.getcwd()
    infile = args.input_bam_path or args.infile
    if not infile:
        sys.exit(not p.print_help())
    samples = read_csv(infile, args)          # paths are made absolute before the chdir, as in the reference
    if workdir != cwd:
        mkdir(workdir, logger=logger)
    sites = op.join(os.getcwd(), "sites")
    os.chdir(workdir)
    ref = args.ref
    repo = TREDsRepo(ref=ref, toy=args.toy, sites=sites)
    repo.set_ploidy(args.haploid)
    treds = args.tred or repo.names
    if args.toy:
        treds = ["HD"]
    task_args = []
    for samplekey, bam, tred in samples:
        jsonfile = ".".join((samplekey, "json"))
        if args.checkexists and op.exists(jsonfile):
            continue
        _treds = [tred] if tred else treds
        task_args.append((samplekey, bam, repo, _treds, args.maxinsert, args.fullsearch, args.useclippedreads,
                          (not args.noalts), (not args.norepeatpairs), args.log))
    if not task_args:
        os.chdir(cwd)
        return
    from .engine import Engine
    pool = host_pool(args.cpus, len(task_args))
    engine = Engine(args.gpu)

    def sink(results):
        if not args.no_output:
            write_vcf_json(results, ref, repo, treds, None, quiet=quiet)
    try:
        run_many(task_args, engine, pool=pool, sink=sink)
    finally:
        if pool is not None:
            pool.close()
            pool.join()
    print("Elapsed time={}".format(timedelta(seconds=time.time() - start)), file=sys.stderr)
    os.chdir(cwd)
    if args.cleanup:
        shutil.rmtree(workdir)


if __name__ == '__main__':
    main(sys.argv[1:])
