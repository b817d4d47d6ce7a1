// Launch parameters of the flash attention kernel of libmmgt_hip.so (attention.hip).
#pragma once

namespace {

struct AttnParams {
  const char *q, *k, *v, *k2, *v2;
  char* o;
  long q_bs0, q_bs1, q_ts, k_bs0, k_bs1, k_ts, v_bs0, v_bs1, v_ts, o_bs0, o_bs1, o_ts;
  long k2_bs, k2_ts, v2_bs, v2_ts;
  int bdiv, k2_bdiv, nk2, seg2_first_batch;
  int nq, nk;
  int nqb, npairs, heads;
  float scale_log2e;
  // optional per-row output multiplier (mmgt_attention_scaled): o[b][q][head] *= out_scale[(head / os_heads) * os_gs + b * nq + q]
  const float* out_scale;
  long os_gs;
  int os_heads;
  char* o_twin;       // attn64 only: the output after the LAST tile of key segment 0 also goes here (same strides as o): mmgt_attention_twin
  int heads_inner;    // workgroup order: heads innermost per (batch entry, query block) on one XCD (attention.hip)
};

}  // namespace
