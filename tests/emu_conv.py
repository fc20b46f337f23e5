"""Lane-level numpy emulation of disco_diffdock_amd/csrc/k_conv.hip (conv_fused_kernel) driven by the PACKED
host-side arrays the library exports (ddk_debug_export).  Test infrastructure: lets the CPU-only suite verify
the weight packing, the K-order permutations, the unit tables and the MFMA register-layout bookkeeping
against the oracle without a GPU.  The MFMA layout modelled here is the documented one for
v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D[(r&3)+8(r>>2)+4(l>>5)][l&31]."""
import numpy as np

NS, NV, XW, NE = 24, 6, 84, 72
F_A, F_PV, F_T1O, F_T1E, F_QV, F_C, F_SH, F_STRIDE = 0, 24, 32, 68, 104, 112, 136, 140
OFF_P, OFF_Q, OFF_C = 24, 42, 60
U_R1_S0, U_R1_V, U_T_S, U_T_V, U_PAD = range(5)
LANES = np.arange(64)
EL, HH = LANES & 31, LANES >> 5


def d_row(r, hh):
    return (r & 3) + 8 * (r >> 2) + 4 * hh


def mfma(a, b, Dl):
    """one v_mfma_f32_32x32x2_f32: a,b per-lane scalars [64]; Dl per-lane accumulators [64,16]."""
    A = np.zeros((32, 2), np.float64)
    B = np.zeros((2, 32), np.float64)
    A[EL, HH] = a
    B[HH, EL] = b
    Dfull = A @ B
    out = Dl.copy()
    for r in range(16):
        out[:, r] += Dfull[d_row(r, HH), EL]
    return out


def emulate(ctx, layer, x_pad, src, dst, group_offsets, edge_attr, sh):
    """returns sum[N, XW] (pre-mean) in float64."""
    N = x_pad.shape[0]
    out = np.zeros((N, XW), np.float64)
    units = ctx.export(f'conv.{layer}.units', np.int32).reshape(-1, 4)
    n_tiles = len(units) // 4
    inv_s3, inv_s2 = 1 / np.sqrt(3.0), 1 / np.sqrt(2.0)
    for g in range(4):
        w1p = ctx.export(f'conv.{layer}.w1p.{g}').reshape(3, 9, 64, 4).astype(np.float64)
        b1p = ctx.export(f'conv.{layer}.b1p.{g}').reshape(3, 2, 16).astype(np.float64)
        w2p = ctx.export(f'conv.{layer}.w2p.{g}').reshape(n_tiles, 9, 64, 4).astype(np.float64)
        b2p = ctx.export(f'conv.{layer}.b2p.{g}').reshape(n_tiles, 2, 16).astype(np.float64)
        gb, ge = group_offsets[g], group_offsets[g + 1]
        for e0 in range(gb, ge, 32):
            nvalid = min(32, ge - e0)
            e = e0 + np.minimum(EL, nvalid - 1)
            valid = EL < nvalid
            sn, dn = src[e], dst[e]
            # GEMM1
            kin = np.stack([24 * (s // 12) + 12 * HH + (s % 12) for s in range(36)], 1)   # [64,36]
            bin_ = edge_attr[e[:, None], kin]
            h = np.zeros((64, 36))
            for T in range(3):
                D = b1p[T][HH]                                   # [64,16]
                for s in range(36):
                    D = mfma(w1p[T, s // 4, :, s % 4], bin_[:, s], D)
                if T < 2:
                    h[:, 16 * T:16 * T + 16] = np.maximum(D, 0)
                else:
                    h[:, 32:36] = np.maximum(D[:, :4], 0)
            # F rows (per edge; both halves see the same row)
            F = np.zeros((32, F_STRIDE))
            for i in range(32):
                ee = e[i]
                xr = x_pad[dst[ee]]
                s0, v = sh[ee, 0], sh[ee, 1:4]
                F[i, F_A:F_A + NS] = xr[:NS]
                F[i, F_C:F_C + NS] = xr[OFF_C:OFF_C + NS]
                p = xr[OFF_P:OFF_P + 3 * NV].reshape(NV, 3)
                q = xr[OFF_Q:OFF_Q + 3 * NV].reshape(NV, 3)
                F[i, F_PV:F_PV + NV] = (p @ v) * inv_s3
                F[i, F_QV:F_QV + NV] = (q @ v) * inv_s3
                F[i, F_T1O:F_T1O + 3 * NV] = (p * s0).ravel()
                F[i, F_T1O + 3 * NV:F_T1O + 6 * NV] = (np.cross(q, v[None]) * inv_s2).ravel()
                F[i, F_T1E:F_T1E + 3 * NV] = (np.cross(p, v[None]) * inv_s2).ravel()
                F[i, F_T1E + 3 * NV:F_T1E + 6 * NV] = (q * s0).ravel()
                F[i, F_SH:F_SH + 4] = sh[ee]
            s0l, vl = sh[e, 0], sh[e, 1:4]
            acc = np.zeros((64, 3))
            for t in range(n_tiles):
                D = b2p[t][HH]
                for s in range(36):
                    D = mfma(w2p[t, s // 4, :, s % 4], h[:, s], D)
                for rq in range(4):
                    w0, w1, scale_bits, _ = units[4 * t + rq]
                    kind, flags, ncomp, f_off = w0 & 15, (w0 >> 4) & 15, (w0 >> 8) & 15, w0 >> 16
                    d = D[:, 4 * rq:4 * rq + 4]
                    Fl = F[EL]                                   # per lane its edge's F row
                    if kind == U_T_V:
                        f = Fl[:, f_off:f_off + 12].reshape(64, 4, 3)
                        acc += np.einsum('lrc,lr->lc', f, d)
                    elif kind != U_PAD:
                        part = (Fl[:, f_off:f_off + 4] * d).sum(1)
                        if kind == U_R1_S0:
                            acc[:, 0] += s0l * part
                        elif kind == U_T_S:
                            acc[:, 0] += part
                        else:
                            acc += vl * part[:, None]
                    if flags & 2:
                        scale = np.array([scale_bits], np.int32).view(np.float32)[0]
                        chan = (w1 & 0xffff) + HH * (w1 >> 16)
                        for c in range(ncomp):
                            np.add.at(out, (sn[valid], chan[valid] + c), acc[valid, c] * scale)
                        acc[:] = 0
    return out


def finalize(ctx, layer, summed, deg, x_pad, dout, with_bn=True):
    mean = ctx.export(f'conv.{layer}.bn_mean').astype(np.float64)
    scale = ctx.export(f'conv.{layer}.bn_scale').astype(np.float64)
    bias = ctx.export(f'conv.{layer}.bn_bias').astype(np.float64)
    v = summed / np.maximum(deg, 1)[:, None]
    v = (v - mean) * scale + bias
    v[:, dout:] = 0
    return (v + x_pad)[:, :dout]
