# Address-sanitizer attempt (verdict r5 item 6): the four HIP libraries built with -fsanitize=address for gfx950:xnack+ (the build container's hipcc
# accepts this: ~65 s for libgvd_raster), then the raster GPU tests + the raster stress script under them.  ROCm's device-side ASan needs (a) XNACK
# enabled on the box (HSA_XNACK=1 and a kernel driver that allows it), (b) the host ASan runtime preloaded into the un-instrumented python, and
# (c) for device-side reports the instrumented runtime libraries of /opt/rocm/lib/asan, which this image does NOT ship (ls below).  Whatever the box
# says is logged verbatim; the red-zone guard allocator (r6_guard_all.sh) is the pass that does not depend on any of this.
# libgvd_diffusion is NOT built here: under -fsanitize=address hipcc rejects the hand-written LDS-DMA block of gemm_mfma.hip (the inline asm at
# :257-265, `global_load_lds_dwordx4 %2, %5` with SGPR-constrained operands: "invalid operand for instruction", 135 times, after 24 min 41 s of
# compilation at -O1 in the build container; the same file builds for gfx950:xnack+ WITHOUT the sanitizer in 21 s) -- the instrumentation moves the
# operands out of the scalar registers the instruction needs.  The diffusion kernels' memory safety rests on the guard allocator.
set -u
R=$PWD
A=$R/guidedvd-3dgs_amd/lib/asan
mkdir -p $A
C=$R/guidedvd-3dgs_amd/csrc
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SAN="--offload-arch=gfx950:xnack+ -fsanitize=address -shared-libsan -g -O1 -std=c++17 -fPIC -shared -Wno-unused-function -Wno-pass-failed"
echo "== asan: /opt/rocm/lib/asan: $(ls /opt/rocm/lib/asan 2>&1 | head -3 | tr '\n' ' ')"
echo "== asan: build"
[ -f $A/libgvd_raster.so ] || $HIPCC $SAN -ffp-contract=off -fno-slp-vectorize -o $A/libgvd_raster.so $C/capi.hip $C/raster_forward.hip $C/raster_backward.hip 2>&1 | grep -v "warning\|^ *[0-9]* |\|\^" | tail -5
[ -f $A/libgvd_knn.so ] || $HIPCC $SAN -ffp-contract=off -o $A/libgvd_knn.so $C/knn.hip 2>&1 | tail -3
[ -f $A/libgvd_loss.so ] || $HIPCC $SAN -o $A/libgvd_loss.so $C/ssim.hip 2>&1 | tail -3
ls -la $A
RT=$($HIPCC -print-file-name=libclang_rt.asan-x86_64.so 2>/dev/null)
[ -f "$RT" ] || RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so 2>/dev/null | head -1)
echo "== asan: host runtime $RT"
export HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:verify_asan_link_order=0
echo "== asan: rocminfo xnack: $(/opt/rocm/bin/rocminfo 2>/dev/null | grep -i -m2 'xnack' | tr '\n' ' ')"
echo "== asan: raster tests (ctypes carrier; the compiled operator is not instrumented)"
LD_PRELOAD=$RT GVD_RASTER_LIB=$A/libgvd_raster.so GVD_RASTER_NO_EXT=1 timeout 900 python -m pytest tests/test_raster_gpu.py -m gpu -q -x 2>&1 | grep -v '^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids' | tail -12
echo "== asan: knn + loss tests"
LD_PRELOAD=$RT GVD_KNN_LIB=$A/libgvd_knn.so GVD_LOSS_LIB=$A/libgvd_loss.so timeout 900 python -m pytest tests/test_knn.py tests/test_fused_loss.py -m gpu -q -x 2>&1 | tail -4
echo "== asan: raster stress"
LD_PRELOAD=$RT GVD_RASTER_LIB=$A/libgvd_raster.so GVD_RASTER_NO_EXT=1 timeout 900 python tests/scripts/r5_raster_stress.py 2>&1 | tail -6
