# A/B of s_nop pacing behind attn64's MFMAs (csrc/attn64.hip, ATTN_PACE): builds the variants on the GPU box and times the level-0 launch.
cd $GRAFT_REPO_ROOT/mmgt_amd/csrc
cp ../libmmgt_hip.so /tmp/orig.so
for v in 0 1 2 3 4; do
  /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-result -ffinite-math-only -DATTN_PACE=$v -c attn64.hip -o /tmp/attn64_p$v.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmmgt_hip.so build/gemm.o build/gemm16.o build/ffn.o build/wav2vec.o build/attention.o /tmp/attn64_p$v.o build/tattn.o build/norm.o build/elementwise.o build/smga.o build/conditioning.o
  echo "pace $v: $(cd ../.. && python tools/bench_attn.py 2>/dev/null | head -2 | tr '\n' ' ')"
done
cp /tmp/orig.so ../libmmgt_hip.so
