# A/B of two raster libraries on one box: builds csrc/ of git revision $1 (default HEAD) as lib/libgvd_raster_ab.so next to the working
# tree's libgvd_raster.so and alternates `bench.py --workload raster` between them (GVD_RASTER_LIB).  usage: r5_ab_raster_lib.sh [rev] [rounds]
REV=${1:-HEAD}; N=${2:-3}
D=$(mktemp -d); mkdir -p $D/csrc $D/include
for f in $(git ls-tree --name-only $REV guidedvd-3dgs_amd/csrc/ | grep -E "raster_|capi.hip"); do git show $REV:$f > $D/csrc/$(basename $f); done
git show $REV:include/gvd_raster.h > $D/include/gvd_raster.h
sed -i 's#"../../include/gvd_raster.h"#"'$D'/include/gvd_raster.h"#' $D/csrc/*.hip $D/csrc/*.h 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -I$D/csrc \
  -o guidedvd-3dgs_amd/lib/libgvd_raster_ab.so $D/csrc/raster_forward.hip $D/csrc/raster_backward.hip $D/csrc/capi.hip || exit 1
echo "built $REV -> lib/libgvd_raster_ab.so"
