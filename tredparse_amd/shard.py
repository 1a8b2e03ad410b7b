"""Sample sharding across GPUs -- the reference's only parallelism is a multiprocessing.Pool over
samples (tredparse/tred.py:528-532); here it is one process per GPU with the same partitioning
unit (a sample with all its loci stays on one device, so its BAM is opened once).

No data-path collective exists: sample x locus units are independent.  torch.distributed (RCCL on the
GPUs, gloo in the CPU tests) carries only the barrier and two scalar reductions of the report.
"""


def shard_range(n_samples, rank, world):
    """Contiguous block partition of sample indices: [start, end) of this rank."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_samples, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def sample_owner(sample_index, n_samples, world):
    """Inverse of shard_range."""
    base, extra = divmod(n_samples, world)
    cut = extra * (base + 1)
    if sample_index < cut:
        return sample_index // (base + 1)
    return extra + (sample_index - cut) // base


def balanced_owners(costs, world):
    """Rank of every sample when samples differ in cost (BAM size ~ coverage): longest-processing-time-first --
    samples in order of decreasing cost, each to the rank with the least work so far (ties: lower rank, then input
    order, so the assignment is deterministic).  With equal costs this is round-robin; shard_range's contiguous blocks
    are kept for the benchmark, where samples are alike."""
    if world < 1:
        raise ValueError("bad world")
    load = [0.0] * world
    owner = [0] * len(costs)
    for i in sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i)):
        r = min(range(world), key=lambda r: (load[r], r))
        owner[i] = r
        load[r] += max(float(costs[i]), 1.0)     # (unknown sizes count alike: round-robin)
    return owner


def aggregate(units_local, elapsed_local, dist=None, device=None):
    """Whole-job throughput: (sum of units over ranks) / (max of elapsed over ranks)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return units_local, elapsed_local
    import torch
    t = torch.tensor([elapsed_local], dtype=torch.float64, device=device)
    u = torch.tensor([units_local], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return int(u.item()), float(t.item())


# ---- one process per GPU, started by a parent that never touches a GPU -------------------------------------
def visible_gpus():
    """Number of HIP devices, counted in a short-lived child so that the calling process stays free of any GPU
    runtime state (it is going to start the per-GPU ranks; a process that has initialised the GPU must neither
    fork workers nor be replaced)."""
    import subprocess
    import sys
    # hipGetDeviceCount through ctypes: a child that imports torch for the same number costs 1.5-2.5 s of every
    # command's start (torch is only asked when the HIP runtime cannot be loaded by name)
    code = ("import ctypes\n"
            "n = ctypes.c_int(0)\n"
            "for name in ('libamdhip64.so', '/opt/rocm/lib/libamdhip64.so'):\n"
            "    try:\n"
            "        lib = ctypes.CDLL(name)\n"
            "    except OSError:\n"
            "        continue\n"
            "    print(n.value if lib.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0)\n"
            "    break\n"
            "else:\n"
            "    import torch\n"
            "    print(torch.cuda.device_count())\n")
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        real = max(0, int(out.stdout.strip().splitlines()[-1]))
    except Exception:
        return 0
    return virtual_gpus(real)


def virtual_gpus(real, env=None):
    """TRED_VIRTUAL_GPUS=V: the launcher treats the box's `real` devices as V (rank r's device is r mod V, which is physical
    device (r mod V) mod real) -- the rehearsal of an 8-GPU run on the boxes this was built on, which have one: real
    ranks, real drivers, real gloo, the real NUMA code on the box's sysfs; every record of such a run says `oversubscribed`.
    The physical count is left in TRED_REAL_GPUS for device_entry / spawn_ranks (and the children)."""
    import os
    env = os.environ if env is None else env
    try:
        v = int(env.get("TRED_VIRTUAL_GPUS", "0") or 0)
    except ValueError:
        v = 0
    if v <= 0 or real < 1:
        env.pop("TRED_REAL_GPUS", None)
        return real
    env["TRED_REAL_GPUS"] = str(real)
    return v


def real_gpus(n_devices, env=None):
    """The physical devices behind n_devices counted ones (TRED_VIRTUAL_GPUS)."""
    import os
    env = os.environ if env is None else env
    try:
        r = int(env.get("TRED_REAL_GPUS", "0") or 0)
    except ValueError:
        r = 0
    return r if 0 < r < n_devices else n_devices


def usable_cpus():
    """CPUs this process may really keep busy: the affinity mask, capped by the cgroup CPU quota when there is one
    (a container may see 256 CPUs and be throttled to 16: running 256 busy threads there is slower than 16)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fp:
                fields = fp.read().split()
            if path.endswith("cpu.max"):
                if fields[0] != "max":
                    n = min(n, max(1, int(float(fields[0]) / float(fields[1]) + 0.5)))
            else:
                quota = int(fields[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                        n = min(n, max(1, int(quota / float(fp.read().split()[0]) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def driver_plan(usable, n_gpus, gpu_inflate=True):
    """(driver processes per GPU, scan threads per driver) for a host with `usable` CPUs feeding n_gpus GPUs -- the one
    rule behind tred.py's `--drivers auto` and bench.py's end-to-end leg, a function of the two numbers alone.
    With the BGZF blocks inflated and the pair lengths walked on the GPU, and a sample's JSON / VCF written natively,
    a driver's interpreter costs ~1.5 ms per sample and the device's own front-end time (decode + walks, ~0.8 ms per
    sample) is what bounds the rate: DRIVERS_PER_GPU processes keep the GPU's front end fed (measured on 16 CPUs and one
    GPU, profiles/r05_e2e_grid.txt: 1 x 14 threads 14.1 k genotypes/s, 2 x 8 32.0 k, 3 x 5 37.9 k, 4 x 4 35.0 k,
    6 x 3 34.6 k), the CPUs but one shared among their scan threads.  Host-only, a scan is ~40 ms of a core and a
    driver keeps about five scan threads fed: one driver per five CPUs."""
    g = max(1, n_gpus)
    if not gpu_inflate:
        per_gpu = max(1, usable // (5 * g))
        return per_gpu, max(1, (usable - 1) // (per_gpu * g))
    per_gpu = max(1, min(DRIVERS_PER_GPU, usable // (2 * g)))
    return per_gpu, max(1, min(8, (usable - 1) // (per_gpu * g)))


DRIVERS_PER_GPU = 3


# ---- which host CPUs a rank should live on -------------------------------------------------------------------
# The from-BAM path is host-bound: on a two-socket 8-GPU node a rank whose scan threads and pinned staging
# float across sockets pays the inter-socket link on every inflated byte.  The GPU-less parent reads the topology
# from sysfs (no HIP call: it must stay off the GPU), gives every rank a CPU set on its GPU's NUMA node, and the child
# applies it with sched_setaffinity before anything touches the GPU -- its pinned buffers are allocated afterwards
# and land on that node (first touch).
def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    cpus = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return sorted(set(cpus))


def format_cpulist(cpus):
    cpus = sorted(set(cpus))
    runs, k = [], 0
    while k < len(cpus):
        j = k
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        runs.append(str(cpus[k]) if j == k else "{}-{}".format(cpus[k], cpus[j]))
        k = j + 1
    return ",".join(runs)


def gpu_numa_nodes(root="/"):
    """NUMA node of every GPU in HIP's device order, read from sysfs alone: the KFD topology lists the compute
    nodes in the order the runtime enumerates them (nodes with simd_count > 0 are GPUs), each with the minor of its
    DRM render node, whose PCI device carries numa_node.  [] when the tree is not there; -1 for an unknown node."""
    import os
    base = os.path.join(root, "sys/class/kfd/kfd/topology/nodes")
    try:
        ids = sorted((int(d) for d in os.listdir(base) if d.isdigit()))
    except OSError:
        return []
    nodes = []
    for i in ids:
        props = {}
        try:
            with open(os.path.join(base, str(i), "properties")) as fp:
                for line in fp:
                    key, _, val = line.strip().partition(" ")
                    props[key] = val
        except OSError:
            continue
        if int(props.get("simd_count", "0") or 0) <= 0:
            continue                                   # a CPU node
        node = -1
        try:
            with open(os.path.join(root, "sys/class/drm/renderD{}/device/numa_node".format(int(props["drm_render_minor"])))) as fp:
                node = int(fp.read().strip())
        except (OSError, KeyError, ValueError):
            pass
        nodes.append(node)
    return nodes


def numa_cpus(root="/"):
    """{node: [cpus]} from /sys/devices/system/node/node*/cpulist."""
    import os
    base = os.path.join(root, "sys/devices/system/node")
    out = {}
    try:
        names = os.listdir(base)
    except OSError:
        return out
    for d in names:
        if d.startswith("node") and d[4:].isdigit():
            try:
                with open(os.path.join(base, d, "cpulist")) as fp:
                    out[int(d[4:])] = parse_cpulist(fp.read())
            except (OSError, ValueError):
                pass
    return out


def rank_cpusets(world, n_devices, allowed=None, gpu_nodes=None, node_cpus=None, visible=None):
    """CPU set of every rank (rank r works on device r mod n_devices): the CPUs of its GPU's NUMA node that this
    process may use, cut into disjoint, equal slices among the ranks that share the node; ranks whose GPU's node is
    unknown (or has no allowed CPU) share what is left over the same way; CPUs of nodes without any rank's GPU are
    dealt out to all ranks on top (no allowed CPU stays idle).  `visible`: the HIP_VISIBLE_DEVICES entries
    when they are plain indices (device d of this process is physical device visible[d])."""
    import os
    if allowed is None:
        allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    gpu_nodes = gpu_numa_nodes() if gpu_nodes is None else gpu_nodes
    node_cpus = numa_cpus() if node_cpus is None else node_cpus
    allowed_set = set(allowed)
    node_of = []
    for r in range(world):
        d = r % n_devices if n_devices > 0 else -1
        if d >= 0 and visible and d < len(visible):
            d = visible[d]
        node = gpu_nodes[d] if 0 <= d < len(gpu_nodes) else -1
        if node not in node_cpus or not (allowed_set & set(node_cpus[node])):
            node = -1
        node_of.append(node)
    taken = set()
    for node in set(node_of) - {-1}:
        taken |= allowed_set & set(node_cpus[node])
    pools = {node: sorted(allowed_set & set(node_cpus[node])) for node in set(node_of) - {-1}}
    if -1 in node_of:
        pools[-1] = sorted(allowed_set - taken)
        if not pools[-1]:                  # nothing is left for the ranks without a node: no topology at all, then --
            node_of = [-1] * world         # every rank an equal, disjoint slice of what this process may use
            pools = {-1: sorted(allowed_set)}
    sets = [None] * world
    for node, pool in pools.items():
        members = [r for r in range(world) if node_of[r] == node]
        for k, r in enumerate(members):
            lo, hi = shard_range(len(pool), k, len(members))
            sets[r] = pool[lo:hi] or pool              # (more ranks than CPUs: they share the pool)
    # CPUs of nodes that host no rank's GPU (one GPU in use on a two-socket box, a mask of one socket's GPUs): the
    # from-BAM path is host-bound, and an idle socket costs more than a remote one -- they go to the ranks in equal,
    # disjoint slices on top of the local ones
    idle = sorted(allowed_set - set(c for pool in pools.values() for c in pool))
    for r in range(world if idle else 0):
        lo, hi = shard_range(len(idle), r, world)
        sets[r] = sorted(set(sets[r]) | set(idle[lo:hi]))
    return sets


def apply_rank_cpuset(env=None):
    """In a rank: confine this process to TRED_CPUSET (set by spawn_ranks) -- call it before the first GPU call and
    before any pinned allocation.  Returns the CPUs, or None when there is nothing to apply."""
    import os
    text = (os.environ if env is None else env).get("TRED_CPUSET", "")
    if not text or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = parse_cpulist(text)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    return cpus


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def device_entry(index, env):
    """The HIP_VISIBLE_DEVICES entry that selects this process's `index`-th visible device in a child.  Under an
    inherited mask (HIP_VISIBLE_DEVICES, or CUDA_VISIBLE_DEVICES which the HIP runtime reads in its place) the devices
    a parent counts are the mask's entries, not 0..n-1: a parent confined to "2,3" hands out 2 and 3 -- never someone
    else's device 0.  ROCR_VISIBLE_DEVICES works one layer below (HIP indices are relative to it) and is inherited
    untouched."""
    mask = env.get("HIP_VISIBLE_DEVICES") or env.get("CUDA_VISIBLE_DEVICES") or ""
    entries = [e.strip() for e in mask.split(",") if e.strip()]
    try:
        real = int(env.get("TRED_REAL_GPUS", "0") or 0)      # (TRED_VIRTUAL_GPUS: counted device k is physical k mod real)
    except ValueError:
        real = 0
    if real > 0:
        index %= real
    return entries[index % len(entries)] if entries else str(index)


def rank_env(rank, world, port, device, base=None):
    """Environment of rank `rank`: the torchrun variables (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) plus
    HIP_VISIBLE_DEVICES = its one device (None: leave the visibility alone, e.g. CPU-only tests), so that inside the
    child the device is always index 0.  `device` counts the devices THIS process sees; under an inherited visibility
    mask it is translated through the mask (device_entry).  TRED_RANK_DEVICE records which device that is."""
    import os
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK="0" if device is not None else str(rank), WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TRED_SPAWNED_RANK="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if device is not None:
        entry = device_entry(device, env)
        env["HIP_VISIBLE_DEVICES"] = entry
        env.pop("CUDA_VISIBLE_DEVICES", None)       # one mask only: the two would be applied on top of each other
        env["TRED_RANK_DEVICE"] = entry
    return env


def _visible_indices(env):
    mask = env.get("HIP_VISIBLE_DEVICES") or env.get("CUDA_VISIBLE_DEVICES") or ""
    entries = [e.strip() for e in mask.split(",") if e.strip()]
    return [int(e) for e in entries] if entries and all(e.isdigit() for e in entries) else None


def spawn_ranks(argv, world, n_devices, timeout=None, env=None, stdout=None, cwd=None, cpusets="auto"):
    """Start `world` child processes running `argv` (a full command line), rank r on device r mod n_devices
    (n_devices = 0: no device pinning), wait for all of them and return their exit codes.  Every rank is handed the
    CPUs of its GPU's NUMA node in TRED_CPUSET (rank_cpusets; cpusets=None: leave the affinity alone, or a list).  The children find each
    other through RANK / WORLD_SIZE / MASTER_PORT exactly as under torch.distributed.run; nothing is exchanged
    on the data path (sample x locus units are independent), so no RCCL is involved -- the ranks only meet in a
    barrier and a (sum units, max time) reduction when they want one.
    The reference fans out the same way, one worker per sample (tredparse/tred.py:521-532)."""
    import subprocess
    import os
    port = free_port()
    procs = []
    if cpusets == "auto":
        e = os.environ if env is None else env
        visible = _visible_indices(e)
        real = real_gpus(n_devices, e) if n_devices > 0 else 0
        if 0 < real < n_devices:               # virtual devices: the NUMA node is that of the physical device behind each
            visible = [(visible[k % real] if visible and k % real < len(visible) else k % real) for k in range(n_devices)]
        cpusets = rank_cpusets(world, n_devices, visible=visible) if n_devices > 0 else None
    for r in range(world):
        dev = (r % n_devices) if n_devices > 0 else None
        renv = rank_env(r, world, port, dev, env)
        if cpusets and cpusets[r]:
            renv["TRED_CPUSET"] = format_cpulist(cpusets[r])
        procs.append(subprocess.Popen(argv, env=renv, stdout=stdout, cwd=cwd))
    codes = []
    try:
        for p in procs:
            codes.append(p.wait(timeout=timeout))
    except subprocess.TimeoutExpired:
        for p in procs:
            if p.poll() is None:
                p.kill()          # exactly the processes started here
        raise
    return codes
