#!/usr/bin/env python3
"""Where does the host thread run relative to the GPU?  Prints the device's PCI id, its NUMA node and local CPUs (sysfs),
the process's affinity, then times the hall registration loop (tools/reg_time.py) as the library runs by default (icp_create
narrows the thread to the GPU's node), with ICP_PIN=0 (wherever the scheduler puts it) and pinned to CPUs of another node."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_cpulist(s):
    out = []
    for part in s.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def main():
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    buf = ctypes.create_string_buffer(64)
    rc = hip.hipDeviceGetPCIBusId(buf, 64, 0)
    bus = buf.value.decode().lower()
    print("device 0 pci", bus, "rc", rc)
    base = f"/sys/bus/pci/devices/{bus}"
    node = open(base + "/numa_node").read().strip() if os.path.exists(base + "/numa_node") else "?"
    local = open(base + "/local_cpulist").read().strip() if os.path.exists(base + "/local_cpulist") else ""
    print("numa_node", node, "local_cpulist", local)
    aff = sorted(os.sched_getaffinity(0))
    print("process affinity:", len(aff), "cpus", aff[:4], "...", aff[-4:])
    for n in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
        print(n, open(n).read().strip())
    loc = [c for c in parse_cpulist(local) if c in aff]
    far = [c for c in aff if c not in set(parse_cpulist(local))]
    runs = [("default", None), ("ICP_PIN=0", None), ("far", far[:8] or None)]
    for name, cpus in runs * 2:
        if name == "far" and not cpus:
            print(name, ": no such cpus in the affinity mask")
            continue
        cmd = [sys.executable, os.path.join(ROOT, "tools", "reg_time.py"), "3000"]
        pre = (lambda: os.sched_setaffinity(0, cpus)) if cpus else None
        env = dict(os.environ, ICP_PIN="0") if name != "default" else dict(os.environ)
        out = subprocess.run(cmd, capture_output=True, text=True, preexec_fn=pre, env=env)
        print(f"{name:9s}", (out.stdout.strip().splitlines() or [out.stderr[-200:]])[-1], flush=True)


if __name__ == "__main__":
    main()
