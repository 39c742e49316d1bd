#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.
# Compiles the reference's CPU kernels for this path *in place* from
# /root/reference/csrc/cpu/{cache,attention,pos_encoding}.cpp (no sources are copied) together
# with oracle/ref_binding.cpp into oracle/_ref/libpe_ref.so.  The reference's own
# build system (cmake + generated registration) is not used.  -mavx512bf16 is left
# out on purpose: gcc 11 lacks the intrinsic cpu_types_x86.hpp:579 needs, so the
# reference's generic bf16 path (attention.cpp:69-77) is what gets built.
# Skips silently (exit 0) when the reference tree is absent (e.g. on the GPU box,
# which only uses the prebuilt .so that travels with the snapshot).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${PE_REFERENCE_ROOT:-/root/reference}"
OUT="$HERE/_ref"
if [ ! -d "$REF/csrc/cpu" ]; then
  echo "build_ref: $REF/csrc/cpu not present, skipping"
  exit 0
fi
mkdir -p "$OUT"
TORCH_INC=$(python3 - <<'EOF'
import torch.utils.cpp_extension as e
print(" ".join("-I" + p for p in e.include_paths()))
EOF
)
TORCH_LIB=$(python3 -c "import torch.utils.cpp_extension as e; print(e.library_paths()[0])")
PY_INC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
CXXFLAGS="-std=c++17 -O2 -fPIC -fopenmp -mavx512f -mavx512vl -mavx512bw -mavx512dq -mf16c -DVLLM_CPU_EXTENSION -D_GLIBCXX_USE_CXX11_ABI=1 -w"
newer() { [ ! -f "$2" ] || [ "$1" -nt "$2" ]; }
for f in cache attention pos_encoding; do
  if newer "$REF/csrc/cpu/$f.cpp" "$OUT/$f.o"; then
    g++ $CXXFLAGS $TORCH_INC -I"$PY_INC" -I"$REF/csrc/cpu" -I"$REF/csrc" -c "$REF/csrc/cpu/$f.cpp" -o "$OUT/$f.o" &
  fi
done
if newer "$HERE/ref_binding.cpp" "$OUT/ref_binding.o"; then
  g++ $CXXFLAGS $TORCH_INC -I"$PY_INC" -c "$HERE/ref_binding.cpp" -o "$OUT/ref_binding.o" &
fi
wait
g++ -shared -fopenmp -o "$OUT/libpe_ref.so" "$OUT/cache.o" "$OUT/attention.o" "$OUT/pos_encoding.o" "$OUT/ref_binding.o" \
  -L"$TORCH_LIB" -Wl,-rpath,"$TORCH_LIB" -ltorch -ltorch_cpu -lc10
echo "build_ref: built $OUT/libpe_ref.so"
