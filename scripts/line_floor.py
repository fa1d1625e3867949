"""Round 6: the line-granular FLOOR of the headline call.

For the bench graph (R-MAT scale S, scrambled labels, seed S) and the bench mask (visited density 0.5, seed 4242) this script
re-derives the library's layout RULES (popularity order, long rows >= 64 entries, hub rows
>= 1024 entries on 64 classes, the other long rows on 16, 39 936 LDS-resident codes per class, cold entries as column-sorted tiles,
short rows as column-sorted row tiles) with torch on the GPU and counts, per stream, the 128-byte lines that

  * the layout holds                     ("all lines":   what a kernel that streams everything moves),
  * hold >= 1 entry of an admitted row   ("must touch":  what ANY kernel on this layout must move),
  * an ideal row-pure layout would hold  ("admitted bytes": entries of admitted rows x bytes per entry, no padding),

and for the gathered operand the distinct 128-byte lines per tile, per XCD share and overall.  It also prints the segment-length
histogram of the hot strips and prices row-pure variants of their records (padding bytes against skipped bytes at visited
densities 0 / 0.5 / 0.9).  It is a MODEL of the layouts from their rules, not a dump of the library's arrays: the stream totals it
prints are checked against GrX_Matrix_cache_bytes / the kernels' own byte counts in profiles/r06/line_floor.md.

Usage (GPU box):  python scripts/line_floor.py --scale 24 > gpurun_out/line_floor.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--scale", type=int, default=24)
    p.add_argument("--device", default="cuda")
    args = p.parse_args()
    import torch

    from graphblas_amd import synthetic

    dev = args.device
    S = args.scale
    n = 1 << S
    indptr, col = synthetic.rmat_csr(S, device=dev)
    nnz = int(col.numel())
    rowlen = (indptr[1:] - indptr[:-1])
    out = {"scale": S, "n": n, "nnz": nnz}

    # ---- popularity order: vertices by falling column count, ties by falling row length ------------------------
    colcnt = torch.bincount(col.long(), minlength=n)
    order = torch.argsort(colcnt * (int(rowlen.max().item()) + 1) + rowlen, descending=True, stable=True)
    rank = torch.empty(n, dtype=torch.int64, device=dev)
    rank[order] = torch.arange(n, device=dev)
    del order
    row_of = torch.repeat_interleave(torch.arange(n, device=dev), rowlen)
    code = rank[col.long()]            # the entry's column code in the ordered twin
    rlen_e = rowlen[row_of]            # length of the entry's row
    del col

    LDS = 39936
    LONG, HUB = 64, 1024
    is_long = rlen_e >= LONG
    is_hub = rlen_e >= HUB
    ncls_e = torch.where(is_hub, 64, 16)
    hot = is_long & (code < LDS * ncls_e)
    cold = is_long & ~hot
    short = ~is_long
    out["entries"] = {"short": int(short.sum()), "hub_hot": int((hot & is_hub).sum()), "mid_hot": int((hot & ~is_hub).sum()),
                      "hub_cold": int((cold & is_hub).sum()), "mid_cold": int((cold & ~is_hub).sum())}

    def lines_of(nbytes_before, nbytes_each):
        """128-byte line numbers touched by records laid out back to back (first .. last line of each record)."""
        first = nbytes_before // 128
        last = (nbytes_before + nbytes_each - 1) // 128
        return first, last

    results = {}
    for vis in (0.0, 0.5, 0.9):
        gen = torch.Generator(device=dev)
        gen.manual_seed(4242)
        visited = torch.rand(n, generator=gen, device=dev) < vis
        adm_row = ~visited
        adm_e = adm_row[row_of]
        r = {"admitted_entries": int(adm_e.sum())}

        # ---- hot strips: segments of one (level, class, row), padded to 8 entries, one 24-byte record per lane (dictionary) -----
        line = code >> 5
        segs = {}
        ukey_keep = {}
        for name, sel, ncls in (("hub", hot & is_hub, 64), ("mid", hot & ~is_hub, 16)):
            ln = line[sel]
            g, rr = ln // ncls, ln % ncls
            cls = torch.where((g & 1) == 1, ncls - 1 - rr, rr)
            key = cls * n + row_of[sel]
            ukey, cnt = torch.unique(key, return_counts=True)  # sorted by (class, row): the strips' order
            seg_adm = adm_row[ukey % n]
            lanes = (cnt + 7) // 8
            segs[name] = (cnt, lanes, seg_adm, ukey // n)
            ukey_keep[name] = ukey
            del ln, g, rr, cls, key
        rec_all = rec_touch = rec_adm = slots_all = 0
        variants = {}
        for name, (cnt, lanes, seg_adm, scls) in segs.items():
            # records back to back inside a class; every class starts on a fresh chunk (64 lanes)
            rec0 = torch.cumsum(lanes, 0) - lanes
            f, l = lines_of(rec0 * 24, lanes * 24)
            nlines = int(l.max().item()) + 1
            touched = torch.zeros(nlines + 1, dtype=torch.int32, device=dev)
            fa, la = f[seg_adm], l[seg_adm]
            # (a segment touches lines f..l: difference array)
            touched.index_add_(0, fa, torch.ones_like(fa, dtype=torch.int32))
            touched.index_add_(0, la + 1, -torch.ones_like(la, dtype=torch.int32))
            must = int((torch.cumsum(touched, 0)[:nlines] > 0).sum().item())
            rec_all += nlines * 128
            rec_touch += must * 128
            rec_adm += int(cnt[seg_adm].sum().item()) * 3
            slots_all += int(lanes.sum().item()) * 4
            # segment-length histogram + row-pure variants: records of G bytes granularity (a segment starts on a G-byte boundary)
            hist = torch.bincount(torch.clamp((cnt - 1) // 8, max=63), minlength=64).tolist()
            v = {"segments": int(cnt.numel()), "entries": int(cnt.sum().item()), "mean_len": float(cnt.double().mean().item()),
                 "lanes_hist_1_to_64plus": hist}
            for G in (24, 64, 128, 256):
                # 3 bytes per entry; a line of G bytes holds (G // 24) records of 8 entries when G >= 24 ... model: bytes = ceil(3 cnt / G') G
                # with G' = the payload of a G-byte piece (G minus a 4-byte row slot when G > 24: the slot stream moves into the piece)
                pay = 24 if G == 24 else G - 4
                pieces = (cnt * 3 + pay - 1) // pay
                tot = int(pieces.sum().item()) * G + (int(lanes.sum().item()) * 4 if G == 24 else 0)
                adm = int(pieces[seg_adm].sum().item()) * G + (int(lanes.sum().item()) * 4 if G == 24 else 0)
                v[f"piece_{G}B"] = {"bytes_all": tot, "bytes_admitted_pieces": adm}
            # round-6 record format: 10 entries per 32-byte lane record (4 lanes = 40 entries per 128-byte line); segments of rows with
            # >= T entries are padded to whole lines (row-pure: a masked-out row's lines are never fetched), the others to whole records
            seg_rowlen = rowlen[ukey_keep[name] % n]
            pure = {}
            for T in (0, 1024, 2048, 4096, 8192, 16384, 1 << 40):
                big = seg_rowlen >= T
                lanes10 = (cnt + 9) // 10
                lanes_p = torch.where(big, (lanes10 + 3) // 4 * 4, lanes10)
                all_b = int(lanes_p.sum().item()) * 32
                # touched: pure segments only when admitted; shared segments: model = every line (4 rows per line: 94 % at density 0.5)
                touched = int(lanes_p[big & seg_adm].sum().item()) * 32 + int(lanes_p[~big].sum().item()) * 32
                pure[str(T)] = {"bytes_all": all_b, "bytes_touched_model": touched, "slot_bytes": int(lanes_p.sum().item()) * 4}
            v["epl10_pure_lines_by_row_threshold"] = pure
            variants[name] = v
        r["hstrip"] = {"record_lines_all_bytes": rec_all, "record_lines_must_touch_bytes": rec_touch, "admitted_entry_bytes": rec_adm,
                       "lane_slot_bytes": slots_all, "variants": variants}

        # ---- cold tiles and short-row tiles: column-sorted entries with their row; stream = every line ---------------------------
        def tiles(sel, bytes_per_entry, tile_entries, rows_per_tile, label):
            if int(sel.sum().item()) == 0:
                return {"label": label, "entries": 0}
            rows_sel = row_of[sel]
            codes_sel = code[sel]
            adm_sel = adm_e[sel]
            # tile = consecutive rows (ordered row number = rank of the row) cut at rows_per_tile rows / tile_entries entries
            rrank = rank[rows_sel]
            o = torch.argsort(rrank, stable=True)
            rrank, codes_sel, adm_sel = rrank[o], codes_sel[o], adm_sel[o]
            pos = torch.arange(rrank.numel(), device=dev)
            # (the library cuts by "entries before / E + rows before / R"; live rows only)
            urow, inv = torch.unique_consecutive(rrank, return_inverse=True)
            tile = pos // tile_entries + inv // rows_per_tile
            tile = torch.unique_consecutive(tile, return_inverse=True)[1]
            ntile = int(tile.max().item()) + 1
            stream_all = int(rrank.numel()) * bytes_per_entry
            # inside a tile the entries are sorted by column code: admitted and masked-out entries interleave at random
            key = tile * n + codes_sel
            o2 = torch.argsort(key, stable=True)
            adm_sorted = adm_sel[o2]
            epl = 128 // bytes_per_entry if 128 % bytes_per_entry == 0 else None
            per_line = 128.0 / bytes_per_entry
            lidx = (torch.arange(adm_sorted.numel(), device=dev).double() / per_line).long()
            t = torch.zeros(int(lidx.max().item()) + 1, dtype=torch.int32, device=dev)
            t.index_add_(0, lidx[adm_sorted], torch.ones(int(adm_sorted.sum().item()), dtype=torch.int32, device=dev))
            must = int((t > 0).sum().item()) * 128
            # operand lines of the ADMITTED entries: distinct (tile, line) pairs; lines by how many tiles touch them
            ka = (tile[o2][adm_sorted] * (n >> 5) + (codes_sel[o2][adm_sorted] >> 5))
            uk = torch.unique(ka)
            per_tile_lines = int(uk.numel())
            ul, tcnt = torch.unique(uk % (n >> 5), return_counts=True)
            return {"label": label, "tiles": ntile, "entries": int(rrank.numel()), "admitted_entries": int(adm_sel.sum().item()),
                    "stream_all_bytes": stream_all, "stream_must_touch_bytes": must,
                    "stream_admitted_entry_bytes": int(adm_sel.sum().item()) * bytes_per_entry,
                    "operand_lines_distinct": int(ul.numel()), "operand_bytes_once": int(ul.numel()) * 128,
                    "operand_bytes_once_per_xcd": int(torch.clamp(tcnt, max=8).sum().item()) * 128,
                    "operand_bytes_once_per_tile": per_tile_lines * 128,
                    "gathers_alone_on_their_line_in_tile": None}

        r["ctile"] = tiles(cold, 10, 16384, 4096 * 64, "cold entries of the long rows (column code 4 B + value 4 B + row 2 B)")
        r["rtile"] = tiles(short, 7, 32768, 8192, "short rows (column code 4 B + row 2 B + value code 1 B)")
        # ---- the rest of the call: output, mask, row words ----------------------------------------------------------------------
        r["vectors"] = {"w_read_write_bytes": int(adm_row.sum().item()) * 8, "w_must_touch_line_bytes": n * 4 * 2 if vis < 0.97 else None,
                        "mask_bits_bytes": n // 8 * 2}
        results[str(vis)] = r
        del visited, adm_row, adm_e
    out["by_visited_density"] = results
    out["algorithmic_bytes_visited_0.5"] = None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
