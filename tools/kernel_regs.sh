# register / LDS / scratch use of the kernels of one object file: bash tools/kernel_regs.sh ceres_slam_amd/csrc/ssba_kernels.o [name filter]
set -e
o=$(readlink -f $1); t=$(mktemp -d); cp $o $t/x.o; ( cd $t && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o > /dev/null 2>&1 )
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $t/x.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 | awk '/\.group_segment_fixed_size:/{l=$2} /\.name:/{n=$2} /\.private_segment_fixed_size:/{s=$2} /\.vgpr_count:/{print n, "vgpr", $2, "lds", l, "scratch", s}' | grep -E "${2:-.}"
rm -rf $t
