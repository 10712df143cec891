#!/usr/bin/env bash
# TEST INFRASTRUCTURE — builds the reference's own Cython box ops into oracle/_ref/.
#
# Compiles, from the sources where they lie under /root/reference:
#   lib/utils/cython_bbox_3d.pyx  (unmodified)
#   lib/utils/cython_nms_3d.pyx   (numpy-2 compat: np.int_t -> np.int64_t, dtype=np.int -> np.int64,
#                                  applied with sed to a scratch copy in $TMPDIR; semantics identical on LP64)
# Only the two built .so files are written into oracle/_ref/ (git-ignored).  No reference source,
# generated C or bytecode is kept in the repository.  Used ONLY by tests/ and tests/golden/gen_golden.py
# to pin oracle/ against the real reference; never imported by the product path.
set -euo pipefail
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF/lib/utils" ]; then
  echo "build_ref: $REF not present (GPU box?) - using prebuilt oracle/_ref if any" >&2
  exit 0
fi
mkdir -p "$OUT"
SCRATCH="$(mktemp -d)"
trap 'rm -rf "$SCRATCH"' EXIT
cp "$REF/lib/utils/cython_bbox_3d.pyx" "$SCRATCH/cython_bbox_3d.pyx"
sed -e 's/np\.int_t/np.int64_t/g' -e 's/dtype=np\.int)/dtype=np.int64)/g' \
    "$REF/lib/utils/cython_nms_3d.pyx" > "$SCRATCH/cython_nms_3d.pyx"
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
NPINC=$(python3 -c "import numpy; print(numpy.get_include())")
EXT=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
for m in cython_bbox_3d cython_nms_3d; do
  (cd "$SCRATCH" && cython -3 "$m.pyx" -o "$m.c" >/dev/null 2>&1 || cython "$m.pyx" -o "$m.c")
  # -ffp-contract=off: keep the fp32 expression `inter / (vi + vj - inter)` un-fused, as x86-64 gcc -O2 does
  gcc -O2 -fPIC -shared -ffp-contract=off -Wno-cpp -Wno-unused-function \
      -DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION \
      -I"$PYINC" -I"$NPINC" "$SCRATCH/$m.c" -o "$OUT/$m$EXT"
done
echo "build_ref: wrote $(ls "$OUT")"
