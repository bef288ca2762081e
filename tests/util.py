"""Shared helpers for the test-suite: seeded CSR builders and the decoder of the native
packed format (the format specification of DESIGN.md, restated independently of the C++ packer)."""
import numpy as np

ALL_LENGTHS = [0, 1, 2, 3, 4, 5, 6, 7, 9, 12, 16, 17, 20, 33, 60, 64, 100, 255, 256, 257, 300, 700, 1024, 1030, 2500]


def csr_from_lengths(lens, n_cols, seed, values="uniform", dtype=np.float64):
    rng = np.random.default_rng(seed)
    lens = np.asarray(lens, np.int64)
    rp = np.zeros(lens.size + 1, np.int64)
    np.cumsum(lens, out=rp[1:])
    nnz = int(rp[-1])
    ci = rng.integers(0, n_cols, nnz).astype(np.int32)
    if values == "ones":
        v = np.ones(nnz)
    elif values == "f16":
        v = rng.uniform(0.5, 1.5, nnz)
    else:
        v = rng.uniform(-1, 1, nnz)
    return rp.astype(np.int32), ci, v.astype(dtype)


def mixed_matrix(m, n_cols, seed, lengths=None, p=None, values="uniform", dtype=np.float64):
    rng = np.random.default_rng(seed + 1000)
    lengths = ALL_LENGTHS if lengths is None else lengths
    lens = rng.choice(lengths, size=m, p=p)
    lens = np.minimum(lens, n_cols * 4)
    return csr_from_lengths(lens, n_cols, seed, values, dtype)


def pair_heavy_matrix(m, n_cols, seed, values="uniform", dtype=np.float64):
    """many len-1 and len-3 rows so that common_13 > 0 (needs >= 128 of each)"""
    rng = np.random.default_rng(seed + 7)
    lens = rng.choice([1, 3, 1, 3, 2, 4, 0, 8, 30], size=m)
    return csr_from_lengths(lens, n_cols, seed, values, dtype)


def slot_of(g, t):
    """g = 15-int row of Plan.host_array('short_groups')"""
    split, base, grp, off = g[6], (g[7], g[8]), (g[9], g[10]), (g[11], g[12])
    p = 0 if t < split else 1
    u = t - split if p else t
    return base[p] + (u // grp[p]) * 2 * grp[p] + off[p] + u % grp[p] if grp[p] else base[p] + u


def decode_long_cb(plan):
    """the column-blocked long rows of a column-panel parent (plan.hpp struct LongCB): row -> [(col, val)] from the piece streams, walked unit by
    unit as the kernel walks them ({} when the plan has none)"""
    st = plan.stats
    nL = st["lcb_rows"]
    if nL == 0:
        return {}
    cb = st["lcb_col_block"]
    rows, dst, ptr = plan.host_array("lcb_row_id"), plan.host_array("lcb_row_dst"), plan.host_array("lcb_ptr")
    unit = plan.host_array("lcb_unit").reshape(-1, 3)
    lcol, val = plan.host_array("lcb_lcol"), plan.host_array("lcb_val")
    assert rows.size == nL == dst.size and lcol.size == val.size == st["lcb_elems"] == ptr[-1] and unit.shape[0] == st["lcb_units"]
    order = plan.order_rid
    for i in range(nL):                                   # the destination is the row's position in the plan's y order
        assert (dst[i] == rows[i]) if getattr(plan, "y_order", 0) == 1 else (order[dst[i]] == rows[i])
    out = {int(r): [] for r in rows}
    seen = set()
    for c, q0, q1 in unit.tolist():
        for q in range(q0, q1):
            assert q // nL == c and q not in seen
            seen.add(q)
            for e in range(ptr[q], ptr[q + 1]):
                if lcol[e] == 0xFFFF:
                    assert val[e] == 0
                    continue
                out[int(rows[q % nL])].append((c * cb + int(lcol[e]), float(val[e])))
    for q in range(ptr.size - 1):                         # a piece outside every unit is empty
        assert q in seen or ptr[q + 1] == ptr[q]
    return out


def decode_two_phase(plan, x=None):
    """The two-phase form (plan.hpp struct TwoPhase) from its streams alone, walked the way the two kernels walk them: phase 1 per unit
    (column block, CB-major segments) resolves every local column and drops it at dst[segment]; phase 2 per row block reads (value, local row)
    and what phase 1 left there.  Returns dict output position -> (cols[], vals[]) in stored order (pads removed), as decode_plan does
    (positions of the plan's y order).  With x: also checks nothing and returns (rows, y) with y accumulated in float64."""
    st = plan.stats
    SEG = st["tp_seg_elems"]
    cb = st["tp_col_block"]
    S = st["tp_segments"]
    row0, seg0 = plan.host_array("tp_rb_row0"), plan.host_array("tp_rb_seg0")
    unit = plan.host_array("tp_unit").reshape(-1, 3)
    dst, lcol, lrow, val = plan.host_array("tp_dst"), plan.host_array("tp_lcol"), plan.host_array("tp_lrow"), plan.host_array("tp_val")
    assert dst.size == S and lcol.size == S * SEG and lrow.size == S * SEG and val.size == S * SEG
    assert unit.shape[0] == st["tp_units"] and row0.size == st["tp_row_blocks"] + 1 and sorted(dst.tolist()) == list(range(S))
    gcol = np.full(S * SEG, -1, np.int64)              # phase 1: the global column of every RB-major element
    covered = np.zeros(S, bool)
    for c, s0, s1 in unit.tolist():
        assert 0 < s1 - s0 <= 65536 // SEG
        for s in range(s0, s1):
            assert not covered[s]
            covered[s] = True
            gcol[dst[s] * SEG:(dst[s] + 1) * SEG] = c * cb + lcol[s * SEG:(s + 1) * SEG].astype(np.int64)
    assert covered.all()
    out = {}
    for b in range(row0.size - 1):
        rows = row0[b + 1] - row0[b]
        for e in range(seg0[b] * SEG, seg0[b + 1] * SEG):
            if lrow[e] == 0xFFFF:
                assert val[e] == 0
                continue
            assert lrow[e] < rows
            cs, vs = out.setdefault(int(row0[b] + lrow[e]), ([], []))
            cs.append(int(gcol[e]))
            vs.append(float(val[e]))
    return out


def decode_plan(plan):
    """Rebuild, from the native packed arrays alone, the list of (col, val) per permuted slot.
    Returns dict slot -> (cols[], vals[]) in stored order (pads removed)."""
    prec = plan.precision
    K, CH, SR = (4, 64, 128) if prec == 64 else (16, 256, 256)
    st = plan.stats
    out = {}
    if st.get("two_phase"):
        return decode_two_phase(plan)
    # long rows
    lv, lc = plan.host_array("long_val"), plan.host_array("long_cid")
    pp, pd = plan.host_array("piece_ptr"), plan.host_array("piece_dst")
    mp, md = plan.host_array("multi_ptr"), plan.host_array("multi_dst")
    # piece_dst / multi_dst hold final y indices: slots (Y_PERMUTED) or natural row ids (Y_NATURAL)
    natural = getattr(plan, "y_order", 0) == 1
    inv = np.argsort(plan.order_rid) if natural else None
    dmap = plan.host_array("dst_map")
    if natural and dmap.size:     # a column panel: row r is written to dst_map[r] (the parent's slot) instead of r
        inv = inv[np.argsort(dmap)]
    part_owner = {}
    for i in range(md.size):
        for q in range(mp[i], mp[i + 1]):
            part_owner[q] = int(md[i])
    # 16-bit ids of the long pieces (r6, plan.hpp long_cid16): chunk i of piece p has base long_base[piece_c16[2 p] + i]; a narrow piece's columns are also
    # base + u16 offset (0xFFFF = pad) -- decoded from THOSE here, so that a plan whose 16-bit ids disagreed with its 32-bit ones would not decode to the CSR
    pc16 = plan.host_array("piece_c16").reshape(-1, 2)
    assert pc16.shape[0] == pd.size
    have16 = pd.size > 0 and lc.size > 0
    if have16:
        l16, lb = plan.host_array("long_cid16"), plan.host_array("long_base")
        assert l16.size == lc.size and lb.size == sum(-(-(int(pp[p + 1]) - int(pp[p])) // CH) for p in range(pd.size))
    for p in range(pd.size):
        dst = int(pd[p])
        slot = dst if dst >= 0 else part_owner[~dst]
        if natural:
            slot = int(inv[slot])
        c = lc[pp[p]:pp[p + 1]]
        v = lv[pp[p]:pp[p + 1]]
        if have16:
            o = l16[pp[p]:pp[p + 1]].astype(np.int64)
            bases = np.repeat(lb[pc16[p, 0]: pc16[p, 0] + -(-c.size // CH)], CH)[:c.size].astype(np.int64)
            real = c >= 0
            lo = np.array([c[k:k + CH][c[k:k + CH] >= 0].min() if (c[k:k + CH] >= 0).any() else 0 for k in range(0, c.size, CH)])
            assert np.array_equal(lb[pc16[p, 0]: pc16[p, 0] + lo.size], lo)                      # the base is the chunk's smallest column
            spans_fit = all((c[k:k + CH][c[k:k + CH] >= 0].max() - lo[k // CH] <= 65534) if (c[k:k + CH] >= 0).any() else True for k in range(0, c.size, CH))
            assert bool(pc16[p, 1]) == (spans_fit and c.size >= 4 * CH)                             # (pieces of fewer than four whole chunks keep their 32-bit ids: plan.hpp kLong16MinChunks)
            if pc16[p, 1]:
                assert ((o == 0xFFFF) == ~real).all()
                c = np.where(real, bases + o, -1).astype(c.dtype)                               # the columns as the kernel derives them
                assert np.array_equal(c, lc[pp[p]:pp[p + 1]])
            else:
                assert (o == 0).all()
        keep = c >= 0
        cs, vs = out.setdefault(slot, ([], []))
        cs.extend(c[keep].tolist())
        vs.extend(v[keep].tolist())
    # medium rows
    mptr, mv, mc = plan.host_array("med_ptr"), plan.host_array("med_val"), plan.host_array("med_cid")
    VPL = CH // 64

    ip_all = plan.host_array("irr_ptr")
    batch, shot = (4, 8) if prec == 64 else (2, 4)       # the kernel's pipeline batch / one-shot limit (plan.hpp med_npair; for f16 both branches below pair the same chunks)

    def npair_of(b):
        """a block long enough for the kernel's pipeline (chunks + tail steps of its first row > shot) stores its leading
        nc // batch * batch chunks in pairs (plan.hpp med_npair)"""
        nc = int(mptr[b + 1] - mptr[b])
        nt = -(-int(ip_all[b * 16 + 1] - ip_all[b * 16]) // K)
        mode = st["chunk_pairs"]          # 0: nothing paired (windowed plans / option); 1: pipelined + one-shot f16; 2: + tail-less one-shot f64
        if mode == 0:
            return 0
        return nc // batch * batch if nc + nt > shot else (nc & ~1 if prec == 16 or (nt == 0 and mode == 2) else 0)

    def unpair(seg, npair):
        """[pair][lane][2 chunks][VPL] for the first npair chunks of a block's segment (plan.hpp med_elem_index) -> chunk-major"""
        seg = seg.copy()
        if npair:
            seg[:npair * CH] = seg[:npair * CH].reshape(npair // 2, 64, 2, VPL).transpose(0, 2, 1, 3).reshape(-1)
        return seg

    cid16 = bool(st.get("cid16_on"))
    if cid16:
        # u16 offsets from a per-chunk base column (0xFFFF = pad); the block's first n8 positions carry one-byte offsets instead (0xFF = pad,
        # plane med_cid8, [batch][lane][4 chunks]); position q of a block holds its chunk med_korig[q]
        off16, off8 = plan.host_array("med_cid16").astype(np.int64), plan.host_array("med_cid8").astype(np.int64)
        c8p, korig, base = plan.host_array("med_c8ptr").astype(np.int64), plan.host_array("med_korig").astype(np.int64), plan.host_array("med_base").astype(np.int64)
        assert mc.size == 0 and off16.size + off8.size == base.size * CH and korig.size == base.size and st["cid8_chunks"] == c8p[-1]
    mv_out = np.empty_like(mv)
    mc_out = np.full(mv.size, -1, np.int64)
    for b in range(mptr.size - 1):
        c0, nc, npair = int(mptr[b]), int(mptr[b + 1] - mptr[b]), npair_of(b)
        vals = unpair(mv[c0 * CH:(c0 + nc) * CH], npair).reshape(nc, CH)             # by position
        if not cid16:
            cols = unpair(mc[c0 * CH:(c0 + nc) * CH], npair).reshape(nc, CH).astype(np.int64)
            order = np.arange(nc)
        else:
            n8 = int(c8p[b + 1] - c8p[b])
            oneshot = prec == 64 and nc + (int(ip_all[b * 16 + 1] - ip_all[b * 16]) + 3) // 4 <= 8          # plan.hpp med_oneshot64: pairs instead of batches of four
            G = 2 if oneshot else 4
            assert n8 % G == 0 and n8 <= npair and (n8 == 0 or prec == 64)
            narrow = off8[c8p[b] * CH:(c8p[b] + n8) * CH].reshape(n8 // G, 64, G).transpose(0, 2, 1).reshape(n8, CH) if n8 else np.zeros((0, CH), np.int64)
            w0 = (c0 - int(c8p[b])) * CH
            wide = unpair(off16[w0:w0 + (nc - n8) * CH], npair - n8).reshape(nc - n8, CH)
            bs = base[c0:c0 + nc]
            cols = np.concatenate([np.where(narrow == 0xFF, -1, bs[:n8, None] + narrow), np.where(wide == 0xFFFF, -1, bs[n8:, None] + wide)])
            order = korig[c0:c0 + nc]
            assert sorted(order.tolist()) == list(range(nc)) and (order[npair:] == np.arange(npair, nc)).all()
        for q in range(nc):                                                            # position q -> the block's chunk order[q]
            k = c0 + int(order[q])
            mv_out[k * CH:(k + 1) * CH] = vals[q]
            mc_out[k * CH:(k + 1) * CH] = cols[q]
    mv, mc = mv_out, mc_out
    ip_, iv, ic = plan.host_array("irr_ptr"), plan.host_array("irr_val"), plan.host_array("irr_cid")
    nb = mptr.size - 1
    row_block, row_long = ip_.size - 1, st["row_long"] + st.get("med_rows_as_pieces", 0)   # MFMA-block rows only: shorter medium rows are slabs (short_groups), the longest may be pieces
    # windowed mode: medium position -> y index through med_dst (a slot, or a row id when Y_NATURAL)
    med_dst = plan.host_array("med_dst") if st.get("x_window_on") else None
    if med_dst is not None and natural and inv is None:
        inv = np.argsort(plan.order_rid)
    for b in range(nb):
        nc = mptr[b + 1] - mptr[b]
        blk_c = mc[mptr[b] * CH: mptr[b + 1] * CH].reshape(nc, CH)
        blk_v = mv[mptr[b] * CH: mptr[b + 1] * CH].reshape(nc, CH)
        for rr in range(16):
            r = b * 16 + rr
            if prec == 64:      # element (k, row) at k*16 + row
                idx = np.arange(K) * 16 + rr
            else:               # element (kq, row, j) at kq*64 + row*4 + j ; k = 4*kq + j
                idx = (np.arange(K) // 4) * 64 + rr * 4 + np.arange(K) % 4
            c = blk_c[:, idx].reshape(-1)
            v = blk_v[:, idx].reshape(-1)
            if r >= row_block:
                assert (c == -1).all() and (v == 0).all()
                continue
            keep = c >= 0
            # pads only after the row's last regular entry
            if keep.any():
                assert keep[: keep.sum()].all(), "pad before a real entry"
            cs = c[keep].tolist() + ic[ip_[r]:ip_[r + 1]].tolist()
            vs = v[keep].tolist() + iv[ip_[r]:ip_[r + 1]].tolist()
            if med_dst is None:
                slot = row_long + r
            else:
                slot = int(inv[med_dst[r]]) if natural else int(med_dst[r])
            assert slot not in out
            out[slot] = (cs, vs)
    # short rows
    sg = plan.host_array("short_groups").reshape(-1, 15)
    sv, sc = plan.host_array("short_val"), plan.host_array("short_cid")
    for g in sg:
        L, count, tiles, tile0 = int(g[0]), int(g[1]), int(g[2]), int(g[3])
        eoff = int(g[4]) & 0xFFFFFFFF | (int(g[5]) << 32)
        seg = int(g[13])
        for t in range(count):
            if seg:          # wave-segmented layout (plan.hpp short_elem_index): [tile][lane], 16 / L rows per 16 lanes
                per16 = 16 // L
                tile, r = divmod(t, 4 * per16)
                at = [eoff + tile * 64 + (r // per16) * 16 + (r % per16) * L + k for k in range(L)]
            else:
                tile, lr = divmod(t, SR)
                at = [eoff + (tile * L + k) * SR + lr for k in range(L)]
            slot = slot_of(g, t)
            assert slot not in out
            out[slot] = ([int(sc[a]) for a in at], [sv[a] for a in at])
    # row tiles of a column panel: position j of the parent's output order (its slot, or the row id when it writes in row order) -> the row;
    # the row sits among this panel's empty rows, at the slot its own order_rid gives it
    rt_ptr = plan.host_array("rt_ptr")
    if rt_ptr.size:
        rt_start, rt_mask = plan.host_array("rt_start").astype(np.int64), plan.host_array("rt_mask")
        rt_val, rt_cid = plan.host_array("rt_val"), plan.host_array("rt_cid")
        T = st["row_tile_max"]
        assert 0 < T <= 32 and rt_mask.size == rt_ptr.size - 1 == -(-plan.order_rid.size // 64) and rt_start.size == 64 * rt_mask.size and rt_cid.size == rt_val.size == rt_ptr[-1] == st["row_tile_nnz"]
        row_of_pos = np.argsort(dmap) if dmap.size else np.arange(plan.order_rid.size)
        own_slot = np.argsort(plan.order_rid)
        for t in range(rt_mask.size):
            n = int(rt_ptr[t + 1] - rt_ptr[t])
            for i in range(64):
                s0 = int(rt_start[t * 64 + i]); s1 = int(rt_start[t * 64 + i + 1]) if i < 63 else n
                on = (int(rt_mask[t]) >> i) & 1
                assert 0 <= s1 - s0 <= T and (on or s1 == s0) and (i or s0 == 0)
                if s1 > s0:
                    slot = int(own_slot[row_of_pos[t * 64 + i]])
                    assert out.get(slot, ([], [])) == ([], [])        # (the row is one of the panel plan's own empty rows)
                    a = int(rt_ptr[t]) + s0
                    out[slot] = (rt_cid[a:a + s1 - s0].tolist(), rt_val[a:a + s1 - s0].tolist())
    return out
