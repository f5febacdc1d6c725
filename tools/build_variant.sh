# An experiment build of the library (compiles csrc/experiments.h: -DDBGSOM_EXPERIMENTS): tools/build_variant.sh <name> <source.hip> "<extra hipcc flags>"
#   -> exp_build/libdbgsom_<name>.so (git-ignored, travels to the GPU box); run with DBGSOM_LIB=exp_build/libdbgsom_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; flags=$3
mkdir -p exp_build
C=dbgsom_amd/csrc
make -s -C $C > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-unused-const-variable -Wno-unused-variable -Wno-unused-but-set-variable -ffp-contract=on -DDBGSOM_EXPERIMENTS $flags -I$C -c $C/$src.hip -o exp_build/${src}_$name.o
objs=""
for o in api engine bmu bmu_dma filter accumulate smooth stats; do
  if [ $o = $src ]; then objs="$objs exp_build/${src}_$name.o"; else objs="$objs $C/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -no-hip-rt -shared -fPIC -o exp_build/libdbgsom_$name.so $objs
echo exp_build/libdbgsom_$name.so
