"""Lane-level numpy emulation of disco_diffdock_amd/csrc/k_conv.hip (conv_fused_kernel) driven by the PACKED
host-side arrays the library exports (ddk_debug_export).  Test infrastructure: lets the CPU-only suite verify
the weight packing, the K-order permutations, the unit tables and the MFMA register-layout bookkeeping
against the oracle without a GPU.  The MFMA layout modelled here is the documented one for
v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D[(r&3)+8(r>>2)+4(l>>5)][l&31]."""
import numpy as np

NS, NV, XW, NE = 24, 6, 84, 72
F_A, F_C, F_T1O, F_T1E, F_PQ, F_STRIDE = 0, 24, 48, 84, 120, 132
F_T2O, F_T2E, F_STRIDE2 = 132, 156, 180       # l<=2 tensor product of the confidence model (mode 1)
OFF_P, OFF_Q, OFF_C = 24, 42, 60
T_RA, T_RT, T_TV, T_RTS = range(4)
FL_NONE, FL_S, FL_V = range(3)
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


def emulate(ctx, layer, x_pad, src, dst, group_offsets, edge_attr, sh, mode=0, slots=None, split=False):
    """returns sum[N, n_slots, XW] squeezed to [N, XW] when slots is None (pre-mean) in float64.
    mode 1: l<=2 FCTP layer of the confidence model (9 groups, extra F parts); slots[g] = accumulator slot of group g.
    split=True: the SPLIT kernel's GEMM1 (score model, gather mode): edge_attr[:, :24] is the edge embedding, the x[src][:24] / x[dst][:24]
    columns are replaced by the packed per-node terms (conv.<l>.wn / bnp: node_finalize_pre_kernel) read in the accumulator's order."""
    N = x_pad.shape[0]
    n_groups = len(group_offsets) - 1
    n_slots = 1 if slots is None else max(slots) + 1
    out_all = np.zeros((N, n_slots, XW), np.float64)
    tiles = ctx.export(f'conv.{layer}.tiles', np.int32).reshape(-1, 4)
    n_tiles = len(tiles)
    inv_s3, inv_s2 = 1 / np.sqrt(3.0), 1 / np.sqrt(2.0)
    for g in range(n_groups):
        out = out_all[:, 0 if slots is None else slots[g]]
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
            if split:
                wn = ctx.export(f'conv.{layer}.wn').reshape(2, 4, NE, NS).astype(np.float64)
                bnp = ctx.export(f'conv.{layer}.bnp').reshape(2, 4, NE).astype(np.float64)
                recv_type, send_type = [0, 0, 1, 1][g], [0, 1, 1, 0][g]       # ligand atom = 0, residue = 1 (group order ll, lr, rr, rl)
                ps = x_pad[sn][:, :NS] @ wn[recv_type, g & 1].T + bnp[recv_type, g & 1]          # [64, 72] in 'pos' order
                pd = x_pad[dn][:, :NS] @ wn[send_type, 2 + (g >> 1)].T + bnp[send_type, 2 + (g >> 1)]
            for T in range(3):
                if split:
                    D = np.zeros((64, 16))
                    nreg = 16 if T < 2 else 4
                    for r in range(nreg):
                        pos = 36 * HH + 16 * T + r
                        D[:, r] = ps[LANES, pos] + pd[LANES, pos]
                    for s in range(12):
                        D = mfma(w1p[T, s // 4, :, s % 4], bin_[:, s], D)
                    if T < 2:
                        h[:, 16 * T:16 * T + 16] = np.maximum(D, 0)
                    else:
                        h[:, 32:36] = np.maximum(D[:, :4], 0)
                    continue
                D = b1p[T][HH]                                   # [64,16]
                for s in range(36):
                    D = mfma(w1p[T, s // 4, :, s % 4], bin_[:, s], D)
                if T < 2:
                    h[:, 16 * T:16 * T + 16] = np.maximum(D, 0)
                else:
                    h[:, 32:36] = np.maximum(D[:, :4], 0)
            # F rows (per edge; both halves see the same row)
            F = np.zeros((32, F_STRIDE2 if mode == 1 else F_STRIDE))
            for i in range(32):
                ee = e[i]
                xr = x_pad[dst[ee]]
                s0, v = sh[ee, 0], sh[ee, 1:4]
                F[i, F_A:F_A + NS] = xr[:NS]
                F[i, F_C:F_C + NS] = xr[OFF_C:OFF_C + NS]
                p = xr[OFF_P:OFF_P + 3 * NV].reshape(NV, 3)
                q = xr[OFF_Q:OFF_Q + 3 * NV].reshape(NV, 3)
                pv, qv = (p @ v) * inv_s3, (q @ v) * inv_s3            # F_PQ = [pv0..3 | qv0..3 | pv4 pv5 qv4 qv5]
                F[i, F_PQ:F_PQ + 12] = np.concatenate([pv[:4], qv[:4], pv[4:], qv[4:]])
                # vector parts: row r of a 12-row part, component c -> 12*(r//4) + 4*c + r%4
                t1o = np.concatenate([p * s0, np.cross(q, v[None]) * inv_s2])      # [2nv, 3]
                t1e = np.concatenate([np.cross(p, v[None]) * inv_s2, q * s0])
                for r in range(2 * NV):
                    for c in range(3):
                        F[i, F_T1O + 12 * (r // 4) + 4 * c + r % 4] = t1o[r, c]
                        F[i, F_T1E + 12 * (r // 4) + 4 * c + r % 4] = t1e[r, c]
                if mode == 1:      # (v^ v^T - I/3) p with v^ = sh[1:4]/sqrt3
                    vh = v / np.sqrt(3.0)
                    for base, blk in ((F_T2O, p), (F_T2E, q)):
                        t2 = vh[None] * (blk @ vh)[:, None] - blk * (vh @ vh) / 3.0     # |v^| = 0 for zero-length edges
                        for r in range(NV):
                            for c in range(3):
                                F[i, base + 12 * (r // 4) + 4 * c + r % 4] = t2[r, c]
            s0l, vl = sh[e, 0], sh[e, 1:4]
            accA = np.zeros((64, 4))
            accV = np.zeros((64, 4, 3))
            Fl = np.concatenate([F, np.zeros((32, 16))], 1)[EL]      # per lane its edge's F row (+ the kernel's 16-float pad)
            for t in range(n_tiles):
                D = b2p[t][HH]
                for s in range(36):
                    D = mfma(w2p[t, s // 4, :, s % 4], h[:, s], D)
                w0, chan0 = tiles[t, 0], tiles[t, 1]
                kind, fl, nrq, f_off = w0 & 3, (w0 >> 2) & 3, (w0 >> 4) & 7, w0 >> 16
                d = D.reshape(64, 4, 4)                              # [lane, rq, j]
                if kind == T_TV:
                    f = Fl[:, f_off:f_off + 12].reshape(64, 3, 4)    # [lane, c, j]
                    accV += np.einsum('lcj,lqj->lqc', f, d)
                elif kind == T_RTS:      # shared tail [pv4 pv5 | qv4 qv5]: rows 0,1 close the column that is flushed below, rows 2,3 open the next
                    accV[:, :, 0] += np.einsum('lj,lqj->lq', Fl[:, f_off:f_off + 2], d[:, :, :2])
                else:
                    part = np.einsum('lj,lqj->lq', Fl[:, f_off:f_off + 4], d)
                    if kind == T_RA:
                        accA += part
                    else:
                        accV[:, :, 0] += part
                if w0 & 0x80:        # 6-channel column: accumulator quad 3 carries another a / c row quad for channel pair xp
                    xp, xoff = (w0 >> 8) & 3, (w0 >> 8) & 0x3c
                    accA[:, xp] += np.einsum('lj,lj->l', Fl[:, xoff:xoff + 4], d[:, 3, :])
                if fl:
                    for rq in range(nrq):
                        if fl == FL_S:
                            chan = chan0 + 2 * rq + HH
                            np.add.at(out, (sn[valid], chan[valid]), (accA[:, rq] * s0l + accV[:, rq, 0])[valid])
                        else:
                            chan = chan0 + 3 * (2 * rq + HH)
                            for c in range(3):
                                np.add.at(out, (sn[valid], chan[valid] + c), (accA[:, rq] * vl[:, c] + accV[:, rq, c])[valid])
                    accA[:] = 0
                    accV[:] = 0
                    if kind == T_RTS:
                        accV[:, :, 0] = np.einsum('lj,lqj->lq', Fl[:, f_off + 2:f_off + 4], d[:, :, 2:])
    return out_all[:, 0] if slots is None else out_all


def finalize(ctx, layer, summed, deg, x_pad, dout, with_bn=True):
    mean = ctx.export(f'conv.{layer}.bn_mean').astype(np.float64)
    scale = ctx.export(f'conv.{layer}.bn_scale').astype(np.float64)
    bias = ctx.export(f'conv.{layer}.bn_bias').astype(np.float64)
    v = summed / np.maximum(deg, 1)[:, None]
    v = (v - mean) * scale + bias
    v[:, dout:] = 0
    return (v + x_pad)[:, :dout]
