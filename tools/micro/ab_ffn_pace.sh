# A/B of the s_nop pacing behind role B's MFMAs in csrc/ffn.hip: builds libmmgt_hip_paceN.so variants on the GPU box and times each.
cd $GRAFT_REPO_ROOT/mmgt_amd/csrc
for v in 0 1 2 3 4; do
  /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-result -DMMGT_FFN_PACE=$v -c ffn.hip -o /tmp/ffn_p$v.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmmgt_p$v.so build/gemm.o build/gemm16.o /tmp/ffn_p$v.o build/wav2vec.o build/attention.o build/attn64.o build/tattn.o build/norm.o build/elementwise.o build/smga.o build/conditioning.o
  cp /tmp/libmmgt_p$v.so ../libmmgt_hip.so
  echo "pace $v: $(cd ../.. && python tools/bench_ffn.py 2>/dev/null | sed -n 3p)"
done
