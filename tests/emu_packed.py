"""TEST INFRASTRUCTURE: a numpy model of what the HIP conv kernels compute FROM THE PACKED
IMAGE (tf2_amd/csrc/weight_pack.cpp layouts), used on the CPU to check the packing logic
(exponent windows, Horner shifts, slab lists, kinfo gather, [x|xneg] image layout) against
the oracle before anything runs on a GPU.  It mirrors conv_mfma2.hip / conv_shift.hip's
data flow, not their scheduling."""
import numpy as np

HDR = np.dtype([("magic", "<u4"), ("version", "<u4"), ("n_layers", "<u4"), ("dir_bytes", "<u4"),
                ("total_bytes", "<u8"), ("tables_hash", "<u8"), ("zero_off", "<u8")])
PL = np.dtype([(n, "<i4") for n in ("kind", "TM", "n_mtiles", "n_phases", "nslab", "Np", "signed_in", "Cp_in",
                                     "max_shift", "n_entries", "n_cchunk", "max_ent", "fast", "dual", "fuse_next", "fused_into", "w_share", "w_main_TM")] +
              [(n, "<u8") for n in ("off_w", "off_w2", "off_entries", "off_dir", "off_kinfo", "off_bias",
                                    "off_alpha", "off_beta", "off_lo", "off_dshift", "off_hdr", "hdr_bytes", "off_dbl", "off_pad", "off_unit",
                                    "off_lut", "off_cls")] + [(n, "<i4") for n in ("fc4", "n_cls", "merge_next", "merged_into")])


def parse(blob: np.ndarray):
    h = np.frombuffer(blob[:HDR.itemsize].tobytes(), HDR)[0]
    n = int(h["n_layers"])
    pls = np.frombuffer(blob[HDR.itemsize:HDR.itemsize + n * PL.itemsize].tobytes(), PL)
    # the directory holds n more entries: the layers' wide-tile alternatives (kind 0 where there is none), see parse_alt
    assert int(h["dir_bytes"]) == HDR.itemsize + 2 * n * PL.itemsize
    return h, pls


def parse_alt(blob: np.ndarray):
    h = np.frombuffer(blob[:HDR.itemsize].tobytes(), HDR)[0]
    n = int(h["n_layers"])
    return np.frombuffer(blob[HDR.itemsize + n * PL.itemsize:HDR.itemsize + 2 * n * PL.itemsize].tobytes(), PL)


def i32(blob, off, n):
    return np.frombuffer(blob[off:off + 4 * n].tobytes(), "<i4")


def requant(acc, alpha, beta, relu):
    acc = acc.astype(np.int64)
    t = ((acc * alpha.astype(np.int64)) >> 20).astype(np.int64)
    t = ((t + 2 ** 31) % 2 ** 32 - 2 ** 31)                 # (int) truncation
    t = ((t + beta.astype(np.int64) + 2 ** 31) % 2 ** 32 - 2 ** 31)
    v = ((t >> 14) + 1) >> 1
    v = np.clip(v, -128, 127)
    if relu:
        v = np.maximum(v, 0)
    return v.astype(np.int8)


def nhwc(x_nchw, Cp, signed_half=None):
    """[B,C,H,W] int8 -> [B,H,W,Cp]; with signed_half: [x | (int8)(-x)] halves."""
    B, C, H, W = x_nchw.shape
    t = np.zeros((B, H, W, Cp), np.int8)
    t[..., :C] = np.transpose(x_nchw, (0, 2, 3, 1))
    if signed_half is not None:
        t[..., signed_half:signed_half + C] = (-t[..., :C].astype(np.int16)).astype(np.int8)
    return t


def first_layer_executed(L, img_q):
    """The library's executed form of a 3x3 first layer on a 3-channel image (net.hip Net::init, PrepArgs::rewrite == 2): a POINTWISE
    layer over the im2col image -- per output pixel the 27 values x[c][oh * s - pad + fh][ow * s - pad + fw] in the order c * 9 + fh * 3 + fw
    (zero outside the image) as [x | xneg], 64 bytes.  Returns (LayerSpec of the executed layer, its input tensor [B, OH, OW, 64])."""
    import dataclasses
    B = img_q.shape[0]
    xp = np.zeros((B, 3, L.H + 2 * L.pad_h + 2, L.W + 2 * L.pad_w + 2), np.int8)
    xp[:, :, L.pad_h:L.pad_h + L.H, L.pad_w:L.pad_w + L.W] = img_q
    t = np.zeros((B, L.OH, L.OW, 64), np.int8)
    for c in range(3):
        for fh in range(3):
            for fw in range(3):
                v = xp[:, c, fh:fh + L.stride * L.OH:L.stride, fw:fw + L.stride * L.OW:L.stride][:, :L.OH, :L.OW]
                t[..., c * 9 + fh * 3 + fw] = v
    t[..., 32:59] = (-t[..., :27].astype(np.int16)).astype(np.int8)
    Le = dataclasses.replace(L, C=27, k=1, stride=1, pad_h=0, pad_w=0, H=L.OH, W=L.OW)
    return Le, t


def fc4_tiles(blob, pl, nt):
    """PackLayer::fc4 (weight_pack.cpp): nibble tiles [m-tile * nslab + slab][TM rows][32 bytes] + per-row tables + K classes -> the int8
    window tiles [entry][window][TM][64] the kernel feeds its MFMAs, following conv_fc.hip's fc4_expand byte for byte: a row's 32 bytes
    = [K half h][K step ks] 8 bytes each; word w, byte j: low nibble = K position ks * 32 + h * 16 + 8 w + j, high nibble = that + 4;
    value = table[class of the K position][window][e] with the sign bit (8) negating it."""
    TM, nslab, nm, Np = int(pl["TM"]), int(pl["nslab"]), int(pl["n_mtiles"]), int(pl["Np"])
    nib = np.frombuffer(blob[int(pl["off_w"]):int(pl["off_w"]) + nm * nslab * TM * 32].tobytes(), np.uint8).reshape(nm, nslab, TM, 2, 2, 2, 4)   # [mt][sl][r][h][ks][w][j]
    lut = np.frombuffer(blob[int(pl["off_lut"]):int(pl["off_lut"]) + Np * 32].tobytes(), np.uint8).reshape(nm, TM, 2, 2, 8).astype(np.int16)      # [mt][r][class][window][e]
    cls = (np.frombuffer(blob[int(pl["off_cls"]):int(pl["off_cls"]) + nslab * 64].tobytes(), np.uint8) != 0).astype(np.int64).reshape(nslab, 64)
    assert int(pl["n_cls"]) in (1, 2) and (int(pl["n_cls"]) == 2 or not cls.any())
    codes = np.zeros((nm, nslab, TM, 64), np.uint8)
    for h in range(2):
        for ks in range(2):
            for w in range(2):
                for j in range(4):
                    k = ks * 32 + h * 16 + 8 * w + j
                    codes[..., k] = nib[..., h, ks, w, j] & 15
                    codes[..., k + 4] = nib[..., h, ks, w, j] >> 4
    out = np.zeros((nm * nslab, nt, TM, 64), np.int8)
    mt_i = np.arange(nm)[:, None, None, None]; r_i = np.arange(TM)[None, None, :, None]
    cl = np.broadcast_to(cls[None, :, None, :], codes.shape)
    for wdw in range(nt):
        v = lut[mt_i, r_i, cl, wdw, (codes & 7).astype(np.int64)]
        # (the kernel feeds the one's complement v ^ 0xff = -v - 1 for negative weights and adds sum(x over their K positions) through
        #  one more MFMA: together -v, also where the weight is zero in this window)
        v = np.where(codes & 8, (v ^ 0xff).astype(np.uint8).astype(np.int8).astype(np.int16) + 1, v)
        out[:, wdw] = v.reshape(nm * nslab, TM, 64).astype(np.int8)
    return out


def conv_from_packed(blob, pl, L, x_t, res=None):
    """L: LayerSpec.  x_t: input tensor [B,H,W,Cp_in] int8.  Returns conv-stage output NCHW
    [B,N,OH,OW] after requant/relu/residual (before pool / global average)."""
    B, H, W, Cp = x_t.shape
    assert Cp == int(pl["Cp_in"]) or int(pl["kind"]) == 2
    # doubled input channels (weight_pack.cpp): stored as 2x - 128, an out-of-range tap reads the pad row (-128 there)
    pad = None
    if int(pl["off_pad"]):
        pad = np.frombuffer(blob[int(pl["off_pad"]):int(pl["off_pad"]) + Cp].tobytes(), np.int8)
        x_t = x_t.copy()
        d = pad == -128
        assert (x_t[..., d] >= 0).all()
        x_t[..., d] = (2 * x_t[..., d].astype(np.int16) - 128).astype(np.int8)
    N, OH, OW = L.N, L.OH, L.OW
    Np = int(pl["Np"])
    bias = i32(blob, int(pl["off_bias"]), Np).astype(np.int64)
    alpha = i32(blob, int(pl["off_alpha"]), Np)
    beta = i32(blob, int(pl["off_beta"]), Np)
    npix = B * OH * OW
    if int(pl["kind"]) == 1:
        TM, P, nslab, nm = int(pl["TM"]), int(pl["n_phases"]), int(pl["nslab"]), int(pl["n_mtiles"])
        entries = i32(blob, int(pl["off_entries"]), max(1, int(pl["n_entries"])))
        dirs = i32(blob, int(pl["off_dir"]), nm * (P + 1)).reshape(nm, P + 1)
        kinfo = i32(blob, int(pl["off_kinfo"]), nslab * 4).reshape(nslab, 4)
        lo = i32(blob, int(pl["off_lo"]), Np).astype(np.int64)
        dsh = i32(blob, int(pl["off_dshift"]), P * Np).reshape(P, Np).astype(np.int64)
        dual = int(pl["dual"])
        nt = 2 if dual else 1                      # weight tiles per entry
        # weight tiles: the layer's own storage, or (alternative entries, PackLayer::w_share) the main entry's tiles of the other
        # height -- halves of 128-row tiles / pairs of the 64-row tiles of two neighbouring m-tiles (ConvArgs w_* in the kernels)
        sTM = int(pl["w_main_TM"]) if int(pl["w_share"]) else TM
        if int(pl["fc4"]):
            st = fc4_tiles(blob, pl, nt)                # 4-bit codes expanded as conv_fc.hip does (fc4_expand)
        else:
            st = np.frombuffer(blob[int(pl["off_w"]):int(pl["off_w"]) + int(pl["n_entries"]) * nt * sTM * 64].tobytes(), np.int8)
            st = st.reshape(-1, nt, sTM, 64)
        nent0 = int(dirs[0, P] - dirs[0, 0])

        class _Tiles:
            def __getitem__(self, key):
                e, w = key
                if sTM == TM:
                    return st[e, w]
                if sTM == 2 * TM:
                    return st[e, w][(cur_mt[0] & 1) * TM:(cur_mt[0] & 1) * TM + TM]
                return np.concatenate([st[e, w], st[e + nent0, w]], axis=0)
        wt = _Tiles()
        cur_mt = [0]
        # gather all slabs once: Bmat[slab] = [npix, 64]
        pb, poh, pow_ = np.unravel_index(np.arange(npix), (B, OH, OW))
        slabs = {}

        def slab(sl):
            if sl in slabs:
                return slabs[sl]
            m = np.zeros((npix, 64), np.int8)
            for sg in range(4):
                ki = int(kinfo[sl, sg]) & 0xffffffff
                coff = ki & 0xffff
                if coff == 0xffff:
                    continue
                dh = (ki >> 16) & 0xff; dw = ki >> 24
                ih = poh * L.stride - L.pad_h + int(dh); iw = pow_ * L.stride - L.pad_w + dw
                ok = (ih >= 0) & (ih < H) & (iw >= 0) & (iw < W)
                v = np.zeros((npix, 16), np.int8)
                if pad is not None:
                    v[:] = pad[coff:coff + 16]
                v[ok] = x_t[pb[ok], ih[ok], iw[ok], coff:coff + 16]
                m[:, sg * 16:(sg + 1) * 16] = v
            slabs[sl] = m
            return m

        acc = np.zeros((Np, npix), np.int64)
        for mt in range(nm):
            cur_mt[0] = mt
            a = np.zeros((TM, npix), np.int64)
            if dual:
                # both exponent windows per entry: (hi << dshift[1]) + lo, combined once (weight_pack.cpp)
                assert P == 2 and dirs[mt, 1] == dirs[mt, 2]
                lo_acc = np.zeros((TM, npix), np.int64)
                for e in range(dirs[mt, 0], dirs[mt, 2]):
                    sb = slab(int(entries[e])).astype(np.float64).T
                    a = (a + (wt[e, 0].astype(np.float64) @ sb).astype(np.int64)) % 2 ** 32
                    lo_acc = (lo_acc + (wt[e, 1].astype(np.float64) @ sb).astype(np.int64)) % 2 ** 32
                a = ((a << dsh[1, mt * TM:(mt + 1) * TM, None]) + lo_acc) % 2 ** 32
            else:
                for p in range(P):
                    if p >= 1:
                        a = (a << dsh[p, mt * TM:(mt + 1) * TM, None]) % 2 ** 32
                    for e in range(dirs[mt, p], dirs[mt, p + 1]):
                        a = (a + (wt[e, 0].astype(np.float64) @ slab(int(entries[e])).astype(np.float64).T).astype(np.int64)) % 2 ** 32
            acc[mt * TM:(mt + 1) * TM] = a
        acc = (bias[:, None] + (acc << lo[:, None])) % 2 ** 32
    else:
        k, taps, ncc = L.k, L.k * L.k, int(pl["n_cchunk"])
        cnt = (Np // 8) * ncc * taps * 128
        if int(pl["fast"]):
            # packed 4-bit filters (weight_pack.cpp): nibble {sign << 3 | e}, e = 7 zero, shift = A[n] + B[c] - e
            nb = np.frombuffer(blob[int(pl["off_w"]):int(pl["off_w"]) + cnt // 2].tobytes(), np.uint8)
            v = np.stack([nb & 15, nb >> 4], axis=1).reshape(Np // 8, ncc, taps, 2, 8, 8).astype(np.int64)
            ab = np.frombuffer(blob[int(pl["off_w2"]):int(pl["off_w2"]) + Np + ncc * 16].tobytes(), np.int8).astype(np.int64)
            A = ab[:Np].reshape(Np // 8, 1, 1, 1, 8, 1)
            Bc = ab[Np:].reshape(1, ncc, 1, 2, 1, 8)
            e = v & 7
            mag = np.where(e == 7, 0, np.left_shift(np.int64(1), np.clip(A + Bc - e, 0, 31)))
            neg = (v & 8) != 0
            if int(pl["signed_in"]):
                w, w2 = np.where(neg, 0, mag), np.where(neg, mag, 0)
            else:
                w, w2 = np.where(neg, -mag, mag), None
        else:
            w = i32(blob, int(pl["off_w"]), cnt).reshape(Np // 8, ncc, taps, 2, 8, 8).astype(np.int64)
            w2 = i32(blob, int(pl["off_w2"]), cnt).reshape(Np // 8, ncc, taps, 2, 8, 8).astype(np.int64) if int(pl["signed_in"]) else None
        pb, poh, pow_ = np.unravel_index(np.arange(npix), (B, OH, OW))
        acc = np.zeros((Np, npix), np.int64)
        for cc in range(ncc):
            for t in range(taps):
                fh, fw = divmod(t, k)
                ih = poh * L.stride - L.pad_h + fh * L.dil; iw = pow_ * L.stride - L.pad_w + fw * L.dil
                ok = (ih >= 0) & (ih < H) & (iw >= 0) & (iw < W)
                xv = np.zeros((npix, 16), np.int64)
                xv[ok] = x_t[pb[ok], ih[ok], iw[ok], cc * 16:cc * 16 + 16]
                xn = (-xv).astype(np.int8).astype(np.int64)
                for n8 in range(Np // 8):
                    wm = w[n8, cc, t].transpose(1, 0, 2).reshape(8, 16)      # [n][half*8+c]
                    acc[n8 * 8:(n8 + 1) * 8] += wm @ xv.T          # exact: int64
                    if w2 is not None:
                        wm2 = w2[n8, cc, t].transpose(1, 0, 2).reshape(8, 16)
                        acc[n8 * 8:(n8 + 1) * 8] += wm2 @ xn.T
        acc = (bias[:, None] + acc) % 2 ** 32
    acc = ((acc + 2 ** 31) % 2 ** 32 - 2 ** 31)
    y = requant(acc[:N], alpha[:N, None], beta[:N, None], L.relu)          # [N, npix], PHYSICAL row order
    y = y.reshape(N, B, OH, OW).transpose(1, 0, 2, 3)
    if res is not None:
        s = np.clip(y.astype(np.int16) + res.astype(np.int16), -128, 127)
        if L.add_relu:
            s = np.maximum(s, 0)
        y = s.astype(np.int8)
    return np.ascontiguousarray(y)


def conv_stem_from_packed(blob, pl, L, x_nchw):
    """conv_stem.hip's data flow for the executed first layer: ONE copy of x per pixel (32 bytes), signed window values
    [window][tap][K half][64 rows][16] at off_w2, and the x = -128
    correction -2 * w * x128 for the negative weights derived from the same tiles (|w| of the negative bytes, twice)."""
    assert int(pl["kind"]) == 1 and int(pl["off_w2"]) != 0 and int(pl["Np"]) == 64 and L.k == 3
    P = int(pl["n_phases"])
    unit = None
    PS = P
    if int(pl["off_unit"]):
        # the low window is the unit-tap mask: the stem image holds the high window alone (weight_pack.cpp off_unit)
        assert P == 2
        unit = np.frombuffer(blob[int(pl["off_unit"]):int(pl["off_unit"]) + 9 * 32].tobytes(), np.int8).reshape(9, 32).astype(np.int64)
        assert set(np.unique(unit).tolist()) <= {0, 1}
        PS = 1
    B, C, H, W = x_nchw.shape
    N, OH, OW = L.N, L.OH, L.OW
    assert OH == H - 2 and OW == W - 2 and C <= 32
    st = np.frombuffer(blob[int(pl["off_w2"]):int(pl["off_w2"]) + PS * 9 * 64 * 32].tobytes(), np.int8).reshape(PS, 9, 2, 64, 16)
    w = np.ascontiguousarray(st.transpose(0, 1, 3, 2, 4)).reshape(PS, 9, 64, 32).astype(np.int64)
    negmag = np.where(w < 0, -w, 0)
    x_t = np.zeros((B, H, W, 32), np.int64)
    x_t[..., :C] = np.transpose(x_nchw, (0, 2, 3, 1))
    x128 = np.where(x_t == -128, -128, 0)
    npix = B * OH * OW
    win = []
    for p in range(PS):
        a = np.zeros((64, npix), np.int64)
        for t in range(9):
            dh, dw = divmod(t, 3)
            xs = x_t[:, dh:dh + OH, dw:dw + OW, :].reshape(npix, 32)
            qs = x128[:, dh:dh + OH, dw:dw + OW, :].reshape(npix, 32)
            a += w[p, t] @ xs.T + 2 * (negmag[p, t] @ qs.T)
        win.append(a % 2 ** 32)
    if unit is not None:                 # S(pixel): the masked inputs' sum, the same for every output row (conv_stem.hip UNIT)
        s_px = np.zeros(npix, np.int64)
        for t in range(9):
            dh, dw = divmod(t, 3)
            s_px += x_t[:, dh:dh + OH, dw:dw + OW, :].reshape(npix, 32) @ unit[t]
        win.append(np.broadcast_to(s_px[None, :] % 2 ** 32, (64, npix)))
    bias = i32(blob, int(pl["off_bias"]), 64).astype(np.int64)
    alpha = i32(blob, int(pl["off_alpha"]), 64)
    beta = i32(blob, int(pl["off_beta"]), 64)
    lo = i32(blob, int(pl["off_lo"]), 64).astype(np.int64)
    dsh = i32(blob, int(pl["off_dshift"]), P * 64).reshape(P, 64).astype(np.int64)
    acc = win[0]
    for p in range(1, P):
        acc = ((acc << dsh[p][:, None]) + win[p]) % 2 ** 32
    acc = (bias[:, None] + (acc << lo[:, None])) % 2 ** 32
    acc = ((acc + 2 ** 31) % 2 ** 32 - 2 ** 31)
    y = requant(acc[:N], alpha[:N, None], beta[:N, None], L.relu)
    return np.ascontiguousarray(y.reshape(N, B, OH, OW).transpose(1, 0, 2, 3))
