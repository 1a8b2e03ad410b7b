#!/bin/bash
# tools/h2d_probe.sh [TAG] -- GPU box: tools/h2d_probe.hip alone, with the runtime's copy kernels instead of the SDMA engines, three
# processes at once; the PCIe link as sysfs reports it.  gpurun_out/TAG_h2d_probe.txt
cd $GRAFT_REPO_ROOT
TAG=${1:-r06}
O=gpurun_out/${TAG}_h2d_probe.txt
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O2 -o /tmp/h2d_probe tools/h2d_probe.hip || exit 1
{
echo "# link (sysfs, display-class devices):"
for d in /sys/bus/pci/devices/*; do
    c=$(cat $d/class 2>/dev/null)
    case "$c" in 0x03*|0x12*) echo "$(basename $d) class $c speed $(cat $d/current_link_speed 2>/dev/null) width $(cat $d/current_link_width 2>/dev/null) max $(cat $d/max_link_speed 2>/dev/null) x$(cat $d/max_link_width 2>/dev/null)";; esac
done
echo "# one process:"; /tmp/h2d_probe
echo "# one process, HSA_ENABLE_SDMA=0 (copy kernels):"; HSA_ENABLE_SDMA=0 /tmp/h2d_probe
echo "# three processes at once:"
for k in 1 2 3; do /tmp/h2d_probe > /tmp/h2d_$k.txt & done; wait; cat /tmp/h2d_1.txt /tmp/h2d_2.txt /tmp/h2d_3.txt
} > $O 2>&1
cat $O | cut -c1-2500
