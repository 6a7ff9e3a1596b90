#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU side (GPU sanitizers are not available on this pool):
#   - the oracle (oracle/icp_oracle.c), built into build/san/libicp_oracle.so
#   - the product's HOST translation units (host_math.cpp, multi.cpp, rccl_dyn.cpp, synth.c), linked into build/san/libmi355_icp.so
#     together with the normally built HIP objects (device code is not instrumented)
# and runs the CPU test suite against both (MI355_ICP_LIB / MI355_ORACLE_LIB point the loaders at them).
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/build/san
mkdir -p "$S"
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined"
gcc -O1 -g $SAN -ffp-contract=off -fno-fast-math -fopenmp -fPIC -Wall -Wextra -std=c11 -shared -o "$S/libicp_oracle.so" "$R/oracle/icp_oracle.c" -lm
make -C "$R/pose_estimation_amd/csrc" -j8 ../lib/libmi355_icp.so > /dev/null
C=$R/pose_estimation_amd/csrc
B=$R/build/csrc
g++ -std=c++17 -O1 -g $SAN -fPIC -ffp-contract=off -fno-fast-math -Wall -Wextra -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c "$C/host_math.cpp" -o "$S/host_math.o"
g++ -std=c++17 -O1 -g $SAN -fPIC -ffp-contract=off -fno-fast-math -Wall -Wextra -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c "$C/rccl_dyn.cpp" -o "$S/rccl_dyn.o"
g++ -std=c++17 -O1 -g $SAN -fPIC -ffp-contract=off -fno-fast-math -Wall -Wextra -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c "$C/multi.cpp" -o "$S/multi.o"
gcc -std=c11 -O1 -g $SAN -fPIC -ffp-contract=off -fno-fast-math -Wall -Wextra -c "$C/synth.c" -o "$S/synth.o"
hipcc --offload-arch=gfx950 -shared -o "$S/libmi355_icp.so" "$B/icp_kernels.o" "$B/icp_fused_kernels.o" "$B/xtarget_kernels.o" "$B/icp_ref_kernels.o" \
  "$B/upstream_kernels.o" "$B/scene_kernels.o" "$B/normals_kernels.o" "$B/comm.o" "$B/icp_engine.o" "$S/host_math.o" "$S/multi.o" "$S/rccl_dyn.o" "$S/synth.o" -ldl -lpthread -Wl,-rpath,/opt/rocm/lib \
  -L"$(dirname "$(gcc -print-file-name=libasan.so)")" -lasan -lubsan
cd "$R"
# leak checking off: the interpreter itself "leaks" by design; everything else on
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  MI355_ICP_LIB="$S/libmi355_icp.so" MI355_ORACLE_LIB="$S/libicp_oracle.so" python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider "$@"
