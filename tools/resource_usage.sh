#!/bin/bash
# profiles/r04_resource_usage.txt: registers, scratch, occupancy of the pivot kernels as the compiler reports them (no GPU needed)
cd "$(dirname "$0")/../piplib_amd/csrc"
echo "# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage on piplib_amd/csrc/pip_adv_e.hip (the lean bulk kernel,"
echo "# pip_lean_kernel<SC, FULL>), pip_adv_a.hip (pip_advance_kernel, one wave per tableau, <= 128 int64 columns, compile-time row"
echo "# capacity, FULL), pip_adv_d.hip + pip_adv_f.hip (128-bit entries: one, four and sixteen waves per tableau), pip_kernels.hip"
echo "# (pip_lean64_kernel among its kernels) and pip_quast.hip (the device-resident traiter(), both flavours), end of round 4"
echo "# (tools/resource_usage.sh)"
for f in pip_adv_e pip_adv_a pip_adv_d pip_adv_f pip_kernels pip_quast; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Rpass-analysis=kernel-resource-usage -x hip -c $f.hip -o /dev/null 2>&1 |
    grep -E "Function Name|TotalSGPRs|VGPRs:|ScratchSize|Occupancy|SGPRs Spill|LDS Size" | sed 's/.*remark: *//;s/ *\[-Rpass-analysis=kernel-resource-usage\]//' |
    while read -r l; do case "$l" in "Function Name:"*) echo "  $(echo "${l#Function Name: }" | c++filt | sed 's/(.*//;s/^void //')";; *) echo "      $l";; esac; done
done
