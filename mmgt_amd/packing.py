"""Host-side weight repacking for the HIP kernels (done once per load_state_dict; plumbing, no hot-path arithmetic)."""
import torch


def pack_geglu(w, b):
    """GEGLU proj (8C, C): interleave per 32 output channels as [32 h rows | 32 gate rows] so that the GEMM epilogue finds
    h and gate of one output element in the same lane (include/mmgt_hip.h: MMGT_ACT_GEGLU)."""
    n2, k = w.shape
    n = n2 // 2
    assert n % 32 == 0
    wp = w.reshape(2, n // 32, 32, k).permute(1, 0, 2, 3).reshape(n2, k).contiguous()
    bp = None if b is None else b.reshape(2, n // 32, 32).permute(1, 0, 2).reshape(n2).contiguous()
    return wp, bp


def pack_conv3x3(w, cin_pad=None, cout_pad=None):
    """(Cout, Cin, 3, 3) -> [Cout][3][3][Cin] (K = (ky, kx, cin), cin fastest), optionally zero padded."""
    cout, cin = w.shape[:2]
    cin_pad = cin_pad or cin
    cout_pad = cout_pad or cout
    out = torch.zeros((cout_pad, 3, 3, cin_pad), device=w.device, dtype=w.dtype)
    out[:cout, :, :, :cin] = w.permute(0, 2, 3, 1)
    return out.contiguous()


def pack_conv3x3_up2(w):
    """(Cout, Cin, 3, 3) -> [4][Cout][2][2][Cin]: the conv behind a nearest 2x upsampling as four 2 x 2 convs on the STORED image, one per output
    phase (a, b) (image index 2 a + b; `hip.conv3x3(..., upsample=2)`).  Output pixel (2 y + a, 2 x + b) reads upsampled rows 2 y + a - 1 .. + 1, i.e.
    stored rows y + a - 1 and y + a: for a = 0 the taps (ky = 0 | ky = 1, 2) fall on them, for a = 1 (ky = 0, 1 | ky = 2); columns alike.  The weights of
    the taps that fall on one stored pixel are summed (in fp32, then rounded once): 16 instead of 36 multiply-adds per stored pixel and output channel,
    the same sums as upsample -> conv3x3 (the zero padding too: a stored row / column outside the image is outside the upsampled one)."""
    cout, cin = w.shape[:2]
    wf = w if w.dtype == torch.float64 else w.float()
    groups = (((0,), (1, 2)), ((0, 1), (2,)))               # [phase][stored row / column 0, 1] -> the 3 x 3 taps that fall on it
    out = torch.zeros((4, cout, 2, 2, cin), device=w.device, dtype=wf.dtype)
    for a in range(2):
        for b in range(2):
            for ty in range(2):
                for tx in range(2):
                    acc = 0
                    for ky in groups[a][ty]:
                        for kx in groups[b][tx]:
                            acc = acc + wf[:, :, ky, kx]
                    out[2 * a + b, :, ty, tx, :] = acc
    return out.to(w.dtype).contiguous()


def pack_conv_taps(w):
    """(4, 320, 3, 3) -> the pack_rowgemm image of the (64, 320) matrix whose row 4 tap + o is w[o, :, ky, kx] (tap = 3 ky + kx; rows 36 .. 63 zero):
    a 3 x 3 conv with four output channels as ONE GEMM over the pixels' channels for all nine taps + `hip.conv_taps_gather`."""
    co, cin = w.shape[:2]
    assert co == 4 and tuple(w.shape[2:]) == (3, 3)
    m = torch.zeros((64, cin), device=w.device, dtype=w.dtype)
    m[:36] = w.permute(2, 3, 0, 1).reshape(36, cin)
    return pack_rowgemm(m)


def pad_rows(w, n_pad):
    """Zero-pad the output (row) dimension of a [N, K] weight / [N] bias."""
    if w is None or w.shape[0] == n_pad:
        return w
    out = torch.zeros((n_pad,) + tuple(w.shape[1:]), device=w.device, dtype=w.dtype)
    out[: w.shape[0]] = w
    return out


def pad_cols(w, k_pad):
    if w.shape[1] == k_pad:
        return w.contiguous()
    out = torch.zeros((w.shape[0], k_pad), device=w.device, dtype=w.dtype)
    out[:, : w.shape[1]] = w
    return out


def round_up(x, m):
    return (x + m - 1) // m * m


FF_STAGE = 61 * 1024      # bytes of one sub-block (32 hidden channels) of the fused FeedForward weight image (csrc/ffn.hip)


def pack_ff_fused(w1, b1, w2):
    """FeedForward(GEGLU) weights -> the LDS image of csrc/ffn.hip (mmgt_ff_fused), a uint8 tensor of inner / 32 sub-blocks of 61 KiB.
    w1 (2 * inner, C) = ff.net.0.proj.weight (rows [h | gate], `chunk(2, -1)` order), b1 (2 * inner,), w2 (C, inner) = ff.net.2.weight.
    Per sub-block sb (hidden channels 32 sb .. 32 sb + 31), every 1-KiB fragment lane-linear (lane l = (r = l & 31, hh = l >> 5) owns
    bytes 16 l .. 16 l + 15):
      [ks = 0 .. C/16 - 1][t = h, gate]   8 bf16  w1[t * inner + 32 sb + r][16 ks + 8 hh + j]                       (ff1 A fragments)
      [u = 0 .. C/32 - 1][s = 0, 1]       8 bf16  w2[32 u + r][32 sb + 16 s + 8 (j >> 2) + 4 hh + (j & 3)]          (ff2 A fragments,
                                          k in the order of the GEGLU'd accumulator registers: ffn.hip header)
      64 fp32                             b1[32 sb + i] (i < 32) | b1[inner + 32 sb + i]"""
    inner, C = w2.shape[1], w2.shape[0]
    assert w1.shape == (2 * inner, C) and b1.shape == (2 * inner,) and inner % 32 == 0 and C % 32 == 0
    dev = w1.device
    nsb, ks_n, nu = inner // 32, C // 16, C // 32
    w1 = w1.to(torch.bfloat16)
    w2 = w2.to(torch.bfloat16)
    lane = torch.arange(64, device=dev)
    r, hh = lane & 31, lane >> 5
    j = torch.arange(8, device=dev)
    sb = torch.arange(nsb, device=dev)
    # ff1: rows (nsb, 2, 64), cols (ks, 64, 8)
    rows1 = (torch.tensor([0, inner], device=dev)[None, :, None] + 32 * sb[:, None, None] + r[None, None, :])          # (nsb, 2, 64)
    cols1 = 16 * torch.arange(ks_n, device=dev)[:, None, None] + 8 * hh[None, :, None] + j[None, None, :]             # (ks, 64, 8)
    img1 = w1[rows1[:, None, :, :, None], cols1[None, :, None, :, :]]                                                 # (nsb, ks, 2, 64, 8)
    # ff2: rows (nu, 64), cols (nsb, 2, 64, 8)
    rows2 = 32 * torch.arange(nu, device=dev)[:, None] + r[None, :]                                                   # (nu, 64)
    kperm = 8 * (j >> 2)[None, :] + 4 * hh[:, None] + (j & 3)[None, :]                                                # (64, 8)
    cols2 = 32 * sb[:, None, None, None] + 16 * torch.arange(2, device=dev)[None, :, None, None] + kperm[None, None]  # (nsb, 2, 64, 8)
    img2 = w2[rows2[None, :, None, :, None], cols2[:, None, :, :, :]]                                                 # (nsb, nu, 2, 64, 8)
    bias = torch.stack([b1[:inner].reshape(nsb, 32), b1[inner:].reshape(nsb, 32)], 1).to(torch.float32)              # (nsb, 2, 32)
    out = torch.zeros((nsb, FF_STAGE), device=dev, dtype=torch.uint8)
    n1, n2 = ks_n * 2 * 1024, nu * 2 * 1024
    assert n1 + n2 + 256 <= FF_STAGE
    out[:, :n1] = img1.contiguous().view(torch.uint8).reshape(nsb, n1)
    out[:, n1:n1 + n2] = img2.contiguous().view(torch.uint8).reshape(nsb, n2)
    out[:, n1 + n2:n1 + n2 + 256] = bias.contiguous().view(torch.uint8).reshape(nsb, 256)
    return out.reshape(-1)


def pack_rowgemm(w):
    """Linear weight(s) (N, 320) -> the fragment-major image of csrc/rowgemm.hip (mmgt_rowgemm320), a uint8 tensor of N / 32 column tiles
    of 20 KiB: [tile nt][k-step ks = 0 .. 19] 1-KiB fragments, lane l = (c = l & 31, hh = l >> 5) owns bytes 16 l .. 16 l + 15 =
    8 bf16 w[32 nt + c][16 ks + 8 hh + j].  Several Linears that share their input are stacked along N before packing (q | k | v)."""
    N, K = w.shape
    assert K == 320 and N % 32 == 0
    img = w.to(torch.bfloat16).reshape(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()   # (nt, ks, hh, c, 8)
    return img.view(torch.uint8).reshape(-1)


def pack_ff_proj_out(w):
    """proj_out weight (320, 320) -> the image of the proj_out tail of csrc/ffn.hip (mmgt_ff_fused_po): [output tile v = 0 .. 9][k-step f =
    0 .. 19] 1-KiB fragments, lane l = (n = l & 31, hh = l >> 5) owns 8 bf16 w[32 v + n][32 (f >> 1) + 16 (f & 1) + 8 (j >> 2) + 4 hh + (j & 3)]
    -- k in the order of the FeedForward's output accumulator registers, which, packed to bf16, are the B operand."""
    N, K = w.shape
    assert N == 320 and K == 320
    dev = w.device
    w = w.to(torch.bfloat16)
    lane = torch.arange(64, device=dev)
    n, hh = lane & 31, lane >> 5
    j = torch.arange(8, device=dev)
    kperm = 8 * (j >> 2)[None, :] + 4 * hh[:, None] + (j & 3)[None, :]                                    # (64, 8)
    rows = 32 * torch.arange(N // 32, device=dev)[:, None] + n[None, :]                                   # (10, 64)
    cols = 16 * torch.arange(K // 16, device=dev)[:, None, None] + kperm[None]                            # (20, 64, 8)
    img = w[rows[:, None, :, None], cols[None, :, :, :]]                                                  # (10, 20, 64, 8)
    return img.contiguous().view(torch.uint8).reshape(-1)


TLEG_HEADS, TLEG_HD, TLEG_C = 8, 40, 320


def pack_tleg(wq, wk, wv, wo):
    """The four (320, 320) weights of a level-0 temporal attention (motion_module.py:294-330 to_q / to_k / to_v / to_out.0) -> the fragment-major
    image of csrc/tleg.hip (mmgt_temporal_leg320), a uint8 tensor.  Fragments are 1 KiB for v_mfma_f32_16x16x32_bf16: lane l = (lm = l & 15,
    lq = l >> 4) owns bytes 16 l .. 16 l + 15 = 8 bf16  W[row(lm)][32 ks + 8 lq + j].
    Heads come in pairs (A, B) = (2 g, 2 g + 1) whose 80 output channels form five 16-row tiles A0 A1 AB2 B0 B1: A0 / A1 / B0 / B1 = the head's
    channels 0 .. 15 / 16 .. 31, AB2 = [A's channels 32 .. 39 | B's channels 32 .. 39].  Per pair, for W in (k, v, q): a "wide" chunk
    [ks = 0 .. 9][A0, A1, AB2] (30 KiB + 2 KiB of zeros), then for W in (k, v, q): a "narrow" chunk [ks][B0, B1] (20 KiB).  Then 10 narrow chunks
    of the out-projection, output columns 32 oc .. + 31:  [ks][nt = 0, 1]  Wo[32 oc + 16 nt + lm][ch(ks, lq, j)] with the reduction in the order
    the kernel packs the attention output: k-step h < 8 = channels 4 lq + j (j < 4) and 16 + 4 lq + j - 4 (j >= 4) of head h; k-steps 8 + u =
    the channels 32 .. 39 of heads 4 u .. 4 u + 3: j < 4: head 4 u (lq < 2) / 4 u + 1 (lq >= 2), j >= 4: heads 4 u + 2 / 4 u + 3, channel
    32 + 4 (lq & 1) + (j & 3)."""
    H, HD, C = TLEG_HEADS, TLEG_HD, TLEG_C
    for w in (wq, wk, wv, wo):
        assert tuple(w.shape) == (C, C)
    dev = wq.device
    lane = torch.arange(64, device=dev)
    lm, lq = lane & 15, lane >> 4
    j = torch.arange(8, device=dev)
    ks = torch.arange(C // 32, device=dev)
    cols = 32 * ks[:, None, None] + 8 * lq[None, :, None] + j[None, None, :]                       # (ks, lane, j)
    chunks = []
    bf = [w.to(torch.bfloat16) for w in (wk, wv, wq)]

    def frags(w, rows):                                                                            # rows (tiles, lane) -> (ks, tiles, lane, j) bytes
        return w[rows[None, :, :, None], cols[:, None, :, :]].contiguous().view(torch.uint8).reshape(-1)

    for g in range(H // 2):
        a0, b0 = HD * 2 * g, HD * (2 * g + 1)
        wide = torch.stack([a0 + lm, a0 + 16 + lm, torch.where(lm < 8, a0 + 32 + lm, b0 + 32 + lm - 8)])      # (3, lane)
        narrow = torch.stack([b0 + lm, b0 + 16 + lm])                                                      # (2, lane)
        for w in bf:
            chunks.append(frags(w, wide))
            chunks.append(torch.zeros(2048, device=dev, dtype=torch.uint8))                        # padded to 32 KiB: 8 whole DMA pieces per wave
        for w in bf:
            chunks.append(frags(w, narrow))
    # out-projection: channel of reduction slot (ks, lq, j)
    ch = torch.empty((C // 32, 64, 8), device=dev, dtype=torch.long)
    jj, lqq = j[None, :], lq[:, None]
    for k in range(H):
        ch[k] = HD * k + torch.where(jj < 4, 4 * lqq + jj, 16 + 4 * lqq + jj - 4)
    for u in range(2):
        head = 4 * u + torch.where(jj < 4, 0, 2) + (lqq >= 2).long()
        ch[8 + u] = HD * head + 32 + 4 * (lqq & 1) + (jj & 3)
    wob = wo.to(torch.bfloat16)
    nt = torch.arange(2, device=dev)
    for oc in range(C // 32):
        rows = 32 * oc + 16 * nt[:, None] + lm[None, :]                                            # (nt, lane)
        img = wob[rows[None, :, :, None], ch[:, None, :, :]]                                       # (ks, nt, lane, j)
        chunks.append(img.contiguous().view(torch.uint8).reshape(-1))
    out = torch.cat(chunks)
    assert out.numel() == 4 * (3 * 32768 + 3 * 20480) + 10 * 20480
    return out


def pack_gnconv(w):
    """A (Cout, Cin, 3, 3) conv weight, (Cin, Cout) = (128, 128), (256, 128), (256, 256) or (128, 64) -> the fragment-major image of
    csrc/gnconv.hip (mmgt_gn_silu_conv3x3), a uint8 tensor: per block of min(Cout, 128) output channels (one launch each), Cin / 128 phases x 18
    half-taps of 128 x block bytes:
    [blk][ph][tap = 3 ky + kx][kh][ks2][ct][lane][8 bf16] = W[128 blk + 16 ct + (lane & 15)][128 ph + 64 kh + 32 ks2 + 8 (lane >> 4) + j][ky][kx] --
    1-KiB fragments of v_mfma_f32_16x16x32_bf16 (lane (lm, lq) owns row lm of the 16-channel tile, reduction slots 8 lq .. 8 lq + 7 of the k-step)."""
    cout, cin = w.shape[0], w.shape[1]
    assert tuple(w.shape) == (cout, cin, 3, 3) and (cin, cout) in ((128, 128), (256, 128), (256, 256), (128, 64))
    dev = w.device
    cl = min(cout, 128)
    lane = torch.arange(64, device=dev)
    rows = 16 * torch.arange(cl // 16, device=dev)[:, None] + (lane & 15)[None, :]                           # (ct, lane)
    kk = torch.arange(4, device=dev)                                                                         # k-steps of 32 of a phase = (kh, ks2)
    cols = 32 * kk[:, None, None] + 8 * (lane >> 4)[None, :, None] + torch.arange(8, device=dev)[None, None, :]   # (k, lane, j)
    wt = w.to(torch.bfloat16).permute(2, 3, 0, 1).reshape(9, cout, cin)                                      # (tap, cout, cin)
    img = torch.stack([torch.stack([wt[:, cl * blk + rows[None, :, :, None], 128 * ph + cols[:, None, :, :]] for ph in range(cin // 128)])
                       for blk in range(cout // cl)])                                                        # (blk, ph, tap, k, ct, lane, j)
    return img.contiguous().view(torch.uint8).reshape(-1)


def pack_rconv(w):
    """A (Cout, Cin, 3, 3) conv weight, Cin % 64 == 0, Cout % 16 == 0 -> the fragment-major image of csrc/rconv.hip (mmgt_gn_silu_conv3x3_unet), a
    uint8 tensor: Cin / 64 phases x 9 taps x 2 k-steps of 32 channels, each Cout / 16 pieces of 1 KiB (a workgroup's block of output channels is a
    contiguous run of pieces of every k-step, whatever its width):
    [ph][tap = 3 ky + kx][ks][ct][lane][8 bf16] = W[16 ct + (lane & 15)][64 ph + 32 ks + 8 (lane >> 4) + j][ky][kx] -- fragments of
    v_mfma_f32_16x16x32_bf16 (lane (lm, lq) owns row lm of the 16-channel tile, reduction slots 8 lq .. 8 lq + 7 of the k-step)."""
    cout, cin = w.shape[0], w.shape[1]
    assert tuple(w.shape) == (cout, cin, 3, 3) and cin % 64 == 0 and cout % 16 == 0
    wt = w.to(torch.bfloat16).permute(2, 3, 0, 1).reshape(9, cout // 16, 16, cin // 64, 2, 4, 8)   # (tap, ct, lm, ph, ks, lq, j)
    img = wt.permute(3, 0, 4, 1, 5, 2, 6)                                                            # (ph, tap, ks, ct, lq, lm, j)
    return img.contiguous().view(torch.uint8).reshape(-1)
