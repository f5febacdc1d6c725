# registers, scratch, occupancy and LDS of every kernel of a source file, from the compiler's own remarks
#   bash tools/resource_usage.sh filter      (CPU only: hipcc cross-compiles)
f=${1:-filter}
cd "$(dirname "$0")/../dbgsom_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=on -Rpass-analysis=kernel-resource-usage \
    -c $f.hip -o /tmp/${f}_ru.o 2>&1 | grep -E "Function Name|    VGPRs:|ScratchSize|Occupancy|VGPRs Spill|LDS Size" | paste - - - - - - |
  sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g;s/remark://g;s/[a-z_./]*\(hip\|h\):[0-9]*:1://g;s/Function Name: _ZN6dbgsom//;s/ScratchSize \[bytes\/lane\]/scratch/;s/Occupancy \[waves\/SIMD\]/occ/;s/VGPRs Spill/spill/;s/LDS Size \[bytes\/block\]/lds/' |
  awk -F'\t' '{n=substr($1,1,70); print n, $2,$3,$4,$5,$6}'
