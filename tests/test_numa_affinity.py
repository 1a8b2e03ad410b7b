"""CPU: which host CPUs a rank lives on (8-GPU readiness that can be proven without the node).  A faked sysfs tree
of a two-socket node with eight GPUs -- KFD topology nodes, DRM render nodes with numa_node, NUMA cpulists -- gives every
rank a CPU set on its GPU's node, disjoint from every other rank's; the launcher hands it over in TRED_CPUSET and the
child applies it before anything else.  (The reference has one Pool over samples and no placement at all,
tredparse/tred.py:521-532.)"""
import json
import os
import subprocess
import sys

from tredparse_amd import shard, tred as tredmod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fake_node(tmp_path, gpus=8, sockets=2, cpus_per_socket=64, smt=True):
    """sysfs of a node: `sockets` CPU nodes, `gpus` GPUs spread evenly over them; SMT siblings follow all first threads
    (cpulist 0-63,128-191 for node 0), as EPYC boxes number them."""
    root = tmp_path / "fakeroot"
    total = sockets * cpus_per_socket
    for n in range(sockets):
        d = root / "sys/devices/system/node/node{}".format(n)
        d.mkdir(parents=True)
        lo = n * cpus_per_socket
        text = "{}-{}".format(lo, lo + cpus_per_socket - 1)
        if smt:
            text += ",{}-{}".format(total + lo, total + lo + cpus_per_socket - 1)
        (d / "cpulist").write_text(text + "\n")
    topo = root / "sys/class/kfd/kfd/topology/nodes"
    for n in range(sockets):                       # the CPU sockets come first in the KFD topology
        d = topo / str(n)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 64\nsimd_count 0\ndrm_render_minor 0\n")
    for g in range(gpus):
        d = topo / str(sockets + g)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {}\n".format(128 + g))
        r = root / "sys/class/drm/renderD{}/device".format(128 + g)
        r.mkdir(parents=True)
        (r / "numa_node").write_text("{}\n".format(g * sockets // gpus))
    return str(root)


def test_cpulist_round_trip():
    assert shard.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert shard.format_cpulist([11, 0, 1, 2, 3, 8, 10]) == "0-3,8,10-11"
    assert shard.parse_cpulist("") == []


def test_eight_ranks_on_two_sockets_get_disjoint_node_local_cpus(tmp_path):
    root = fake_node(tmp_path)
    nodes, cpus = shard.gpu_numa_nodes(root), shard.numa_cpus(root)
    assert nodes == [0, 0, 0, 0, 1, 1, 1, 1]
    assert cpus[0] == list(range(0, 64)) + list(range(128, 192)) and cpus[1] == list(range(64, 128)) + list(range(192, 256))
    for world in (8, 16, 32):                      # one, two and four driver processes per GPU
        sets = shard.rank_cpusets(world, 8, allowed=list(range(256)), gpu_nodes=nodes, node_cpus=cpus)
        assert len(sets) == world and all(sets)
        seen = set()
        for r, cs in enumerate(sets):
            assert not (seen & set(cs)), "rank {} shares a CPU".format(r)
            seen |= set(cs)
            assert set(cs) <= set(cpus[nodes[r % 8]]), "rank {} left its GPU's node".format(r)
            assert len(cs) == 256 // world
        assert seen == set(range(256))


def test_visibility_mask_and_restricted_affinity(tmp_path):
    root = fake_node(tmp_path)
    nodes, cpus = shard.gpu_numa_nodes(root), shard.numa_cpus(root)
    # a parent confined to devices 4-7: its device 0 is physical device 4, on node 1
    sets = shard.rank_cpusets(4, 4, allowed=list(range(256)), gpu_nodes=nodes, node_cpus=cpus, visible=[4, 5, 6, 7])
    # every rank has its 32 CPUs of node 1, and node 0 -- no rank's GPU is there -- is dealt out on top, 32 CPUs each:
    # the from-BAM path is host-bound, no allowed CPU stays idle
    assert all(len(set(cs) & set(cpus[1])) == 32 and len(set(cs) & set(cpus[0])) == 32 for cs in sets)
    assert len(set().union(*map(set, sets))) == 256 and sum(len(cs) for cs in sets) == 256
    # one GPU of eight in use, six driver processes on it: node 0's CPUs in six local slices, node 1's dealt out too
    sets = shard.rank_cpusets(6, 1, allowed=list(range(256)), gpu_nodes=nodes, node_cpus=cpus)
    assert sorted(c for cs in sets for c in cs) == list(range(256))
    assert all(20 <= len(set(cs) & set(cpus[0])) <= 22 and 20 <= len(set(cs) & set(cpus[1])) <= 22 for cs in sets)
    # a container that may only use 16 CPUs of node 0: the node-1 GPUs' ranks cannot be local -- every rank then gets
    # an equal, disjoint slice of what there is
    sets = shard.rank_cpusets(8, 8, allowed=list(range(16)), gpu_nodes=nodes, node_cpus=cpus)
    assert sorted(c for cs in sets for c in cs) == list(range(16)) and all(len(cs) == 2 for cs in sets)
    # no topology at all (this container): the same
    sets = shard.rank_cpusets(3, 1, allowed=[0, 1, 2, 3, 4, 5, 6], gpu_nodes=[], node_cpus={})
    assert sorted(c for cs in sets for c in cs) == list(range(7)) and [len(cs) for cs in sets] == [3, 2, 2]
    # more ranks than CPUs: they share
    sets = shard.rank_cpusets(4, 1, allowed=[0, 1], gpu_nodes=[], node_cpus={})
    assert all(cs for cs in sets)


def test_spawned_ranks_apply_their_cpuset_before_anything_else(tmp_path):
    """Two real child processes: each finds TRED_CPUSET, applies it, and reports the affinity it then runs with."""
    have = sorted(os.sched_getaffinity(0))
    if len(have) < 2:
        import pytest
        pytest.skip("needs two CPUs")
    sets = [[have[0]], [have[1]]]
    code = ("import json, os, sys; sys.path.insert(0, {!r}); from tredparse_amd import shard; got = shard.apply_rank_cpuset(); "
            "json.dump(dict(rank=int(os.environ['RANK']), got=got, now=sorted(os.sched_getaffinity(0))), "
            "open(os.path.join({!r}, 'r' + os.environ['RANK'] + '.json'), 'w'))").format(ROOT, str(tmp_path))
    codes = shard.spawn_ranks([sys.executable, "-c", code], 2, 2, timeout=120, cpusets=sets)
    assert codes == [0, 0]
    for r in range(2):
        rec = json.load(open(tmp_path / "r{}.json".format(r)))
        assert rec["got"] == sets[r] and rec["now"] == sets[r]


def test_gpus_8_hands_every_child_its_share(tmp_path, monkeypatch):
    """tred.py --gpus 8 --cpus T: the parent starts eight children, each with the same --cpus T (it is per GPU), its
    one device and -- through spawn_ranks -- the CPUs of that device's NUMA node; without --cpus a child takes the
    usable CPUs divided by the GPUs, capped by its CPU set."""
    started = []

    class P(object):
        def __init__(self, argv, env=None, stdout=None, cwd=None):
            started.append((list(argv), dict(env)))

        def wait(self, timeout=None):
            return 0

        def poll(self):
            return 0

    root = fake_node(tmp_path)
    nodes, cpus = shard.gpu_numa_nodes(root), shard.numa_cpus(root)
    real = shard.rank_cpusets
    monkeypatch.setattr(shard, "rank_cpusets", lambda world, n_devices, **kw: real(world, n_devices, allowed=list(range(256)),
                                                                                   gpu_nodes=nodes, node_cpus=cpus))
    monkeypatch.setattr(subprocess, "Popen", P)
    monkeypatch.setattr(shard, "visible_gpus", lambda: 8)
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    monkeypatch.chdir(tmp_path)
    bam = os.path.join(ROOT, "tests", "golden", "bam", "t001.bam")
    (tmp_path / "samples.csv").write_text("".join("k{:02d},{}\n".format(i, bam) for i in range(16)))
    tredmod.main(["samples.csv", "--workdir", str(tmp_path / "w"), "--gpus", "8", "--cpus", "6", "--tred", "HD", "--no-output"], quiet=True)
    assert len(started) == 8
    seen = set()
    for r, (argv, env) in enumerate(started):
        assert argv[argv.index("--cpus") + 1] == "6" and "--task-file" in argv
        assert env["RANK"] == str(r) and env["HIP_VISIBLE_DEVICES"] == str(r)
        mine = set(shard.parse_cpulist(env["TRED_CPUSET"]))
        assert len(mine) == 32 and not (mine & seen) and mine <= set(cpus[nodes[r]])
        seen |= mine
    # the default of --cpus: all usable CPUs for one GPU (the reference: cpu_count()), an eighth of them per rank of
    # eight, and never more than the rank's CPU set holds
    monkeypatch.setattr(shard, "usable_cpus", lambda: 16)
    assert tredmod.default_cpus(1, None) == 16 and tredmod.default_cpus(8, None) == 2 and tredmod.default_cpus(2, [3]) == 1
    monkeypatch.setattr(shard, "usable_cpus", lambda: 256)
    assert tredmod.default_cpus(8, list(range(32))) == 32 and tredmod.default_cpus(8, list(range(8))) == 8
