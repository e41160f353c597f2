# usage: bash tools/mkvariant.sh <name> "<extra hipcc flags>"  — builds rrt_amd/variant_<name>.so with the flags, then restores the default build
set -e
name=$1; flags=$2
make -s >/dev/null
cp rrt_amd/librrt_amd.so /tmp/rrt_default.so
touch rrt_amd/csrc/kernels.hip
make -s DEVFLAGS="--offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize $flags" >/dev/null
cp rrt_amd/librrt_amd.so rrt_amd/variant_$name.so
cp /tmp/rrt_default.so rrt_amd/librrt_amd.so
touch rrt_amd/csrc/kernels.hip
make -s >/dev/null
echo built rrt_amd/variant_$name.so
