#!/bin/bash
# Rebuilds a PREVIOUS round's kernel library from git for a same-box A/B under the current Python (tools/ab_libraries.sh):
#   bash tools/build_prev_round_library.sh ce9590d r04      -> tools/hip/variants/libdm_r04.so
# The old sources are given the current ABI number and stubs for the entry points added since (they are never called by bench.py).
set -euo pipefail
REV=${1:?git revision}; TAG=${2:?tag}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" deepmerge_amd/csrc include | tar -x -C "$TMP"
CUR=$(grep -o 'return [0-9]*; }' "$ROOT/deepmerge_amd/csrc/dm_api.cpp" | head -1)
sed -i "s/extern \"C\" int dm_abi_version(void) { return [0-9]*; }/extern \"C\" int dm_abi_version(void) { $CUR/" "$TMP/deepmerge_amd/csrc/dm_api.cpp"
if ! grep -q dm_pair_batch_gather "$TMP/include/deepmerge_hip.h"; then
cat >> "$TMP/deepmerge_amd/csrc/dm_api.cpp" <<'EOS'
extern "C" int dm_pair_batch_gather(const void *, int, int, int, int, const void *, const void *, const void *, const void *, int, int, int, int, int, int, void *, int,
                                    const void *, void *, void *, void *) { return -6; }
EOS
fi
if ! grep -q dm_gemm_grouped "$TMP/include/deepmerge_hip.h"; then      # (ABI 6: the separate calls, which is what the entry point means)
cat >> "$TMP/deepmerge_amd/csrc/dm_api.cpp" <<'EOS'
extern "C" int64_t dm_gemm_grouped_workspace_bytes(const DmGemmArgs *, int32_t) { return 0; }
extern "C" int dm_gemm_grouped(const DmGemmArgs *args, int32_t n, void *, int64_t, void *stream) {
  for (int i = 0; i < n; ++i) { const int rc = dm_gemm(&args[i], stream); if (rc != 0) return rc; }
  return 0;
}
EOS
fi
if ! grep -q dm_adam_step_dev_pair "$TMP/include/deepmerge_hip.h"; then      # (bf16x3 runs against such a library: DM_X3_PAIR_MIRROR=0)
cat >> "$TMP/deepmerge_amd/csrc/dm_api.cpp" <<'EOS'
extern "C" int dm_adam_step_dev_pair(float *, const float *, float *, float *, void *, void *, int64_t, const float *, double, double, double, double, void *) { return -6; }
EOS
fi
make -C "$TMP/deepmerge_amd/csrc" -j6 > "$TMP/build.log" 2>&1 || { tail -20 "$TMP/build.log"; exit 1; }
mkdir -p "$ROOT/tools/hip/variants"
cp "$TMP/deepmerge_amd/libdeepmerge_hip.so" "$ROOT/tools/hip/variants/libdm_$TAG.so"
rm -rf "$TMP"
echo "built tools/hip/variants/libdm_$TAG.so from $REV"
