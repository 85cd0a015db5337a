"""Workload construction and result verification shared by ``bench.py`` and the headline-shape GPU tests.

Pure torch on the device that holds the tensors (no oracle, no CPU fallback): the imposed-sparsity read lists of
SURVEY.md §8(d) and a sampled-row check of an attention result against a plain fp32 torch reference that applies the
same block mask. The check is the gate on shapes the CPU oracle cannot finish (S = 75 600, H = 40): it looks at whole
query rows, so a tile that was skipped, walked twice or masked wrongly shows up in the LSE (one 64-key tile of a
43 k-key row moves it by 1.5e-3; the bound is 2e-4) and in O.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence

import torch


# ----------------------------------------------------------------------------------------- imposed lists
def banded_rows(q_tiles: int, k_tiles: int, block_m: int, block_n: int, sparsity: float) -> torch.Tensor:
    """[q_tiles, 5] int32 list-row heads for the imposed-sparsity pattern (<= 2 ranges): every q-tile keeps the first
    walked tile (k_tiles - 1) plus a contiguous band of round((1 - s) * k_tiles) - 1 key tiles centred on its diagonal."""
    keep = max(1, round((1.0 - sparsity) * k_tiles))
    rows = torch.zeros(q_tiles, 5, dtype=torch.int32)
    for m in range(q_tiles):
        if keep >= k_tiles:
            rows[m, :3] = torch.tensor([2, k_tiles - 1, 0])
            continue
        centre = min(k_tiles - 1, (m * block_m + block_m // 2) // block_n)
        band = keep - 1                                   # + the always-walked first tile k_tiles-1
        lo = max(0, min(centre - band // 2, k_tiles - 1 - band))
        hi = lo + band - 1
        if band <= 0:
            rows[m, :3] = torch.tensor([2, k_tiles - 1, k_tiles - 1])
        elif hi >= k_tiles - 2:                           # band touches the first tile: one range
            rows[m, :3] = torch.tensor([2, k_tiles - 1, lo])
        else:
            rows[m] = torch.tensor([4, k_tiles - 1, k_tiles - 1, hi, lo])
    return rows


def listed_tiles_of_rows(rows: torch.Tensor) -> int:
    n = 0
    for r in rows.tolist():
        n += r[1] - r[2] + 1
        if r[0] == 4:
            n += r[3] - r[4] + 1
    return n


def impose_lists(att, rows: torch.Tensor):
    """Overwrite BOTH ping-pong buffers of `att` with the same rows (fixed point under thr=-inf)."""
    sl = att._skip_list
    sl.zero_()
    sl[..., :5] = rows.to(sl.device)[None, None, None]


def executed_flops(rows: torch.Tensor, heads: int, batch: int, S: int, Sk: int, bm: int, bn: int, D: int) -> float:
    """sum over listed tiles of 4*rows*cols*D with ragged edge tiles counted at their real size."""
    k_tiles = -(-Sk // bn)
    total = 0.0
    last_cols = Sk - (k_tiles - 1) * bn
    for m, r in enumerate(rows.tolist()):
        nrows = min(bm, S - m * bm)
        ranges = [(r[1], r[2])] + ([(r[3], r[4])] if r[0] == 4 else [])
        cols = 0
        for s, e in ranges:
            cols += (s - e + 1) * bn
            if s == k_tiles - 1:
                cols -= bn - last_cols
        total += 4.0 * nrows * cols * D
    return total * heads * batch


# ----------------------------------------------------------------------------------------- verification
def listed_key_mask(list_row: Sequence[int], block_n: int, seqlen_k: int, device) -> torch.Tensor:
    """bool[seqlen_k]: keys inside the tiles a skip-list row ``[L, start0, end0, ...]`` lists (ranges descending, both
    ends inclusive; the first range is walked even when L == 0, mainloop_fwd_sm90_tma_gmma_ws.hpp:93-101)."""
    mask = torch.zeros(seqlen_k, dtype=torch.bool, device=device)
    n = max(int(list_row[0]), 2)
    for i in range(1, n, 2):
        s, e = int(list_row[i]), int(list_row[i + 1])
        if s >= e:
            mask[e * block_n: min((s + 1) * block_n, seqlen_k)] = True
    return mask


def sample_rows(seqlen_q: int, n: int, block_m: int, seed: int = 1) -> torch.Tensor:
    """n distinct query rows: the first and last rows of the tensor, one row either side of every q-tile boundary near the
    ends (the zero-padded last q-tile, the first q-tile) and uniformly random rows for the rest."""
    fixed = {0, seqlen_q - 1, min(block_m - 1, seqlen_q - 1), min(block_m, seqlen_q - 1),
             max(0, (seqlen_q - 1) // block_m * block_m - 1), (seqlen_q - 1) // block_m * block_m}
    g = torch.Generator().manual_seed(seed)
    rand = torch.randperm(seqlen_q, generator=g)[: max(0, n)].tolist()
    rows = sorted(fixed | set(rand[: max(0, n - len(fixed))]))
    return torch.tensor(rows, dtype=torch.long)


@torch.no_grad()
def sampled_row_check(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, lse: Optional[torch.Tensor],
                      read_list: Optional[torch.Tensor], block_m: int, block_n: int, heads: Iterable[int],
                      n_rows: int = 256, batch: int = 0, softmax_scale: Optional[float] = None,
                      o_rtol: float = 2.0 ** -8, o_atol: float = 1e-4, lse_atol: float = 2e-4, seed: int = 1) -> Dict:
    """Compare ``n_rows`` sampled query rows of every head in ``heads`` with a plain fp32 torch attention over exactly the
    keys the row's q-tile lists in ``read_list`` ([B, H, Qt, Kt+1]; None = dense).

    q (B,S,H,D) / k, v (B,Sk,Hk,D) bf16 or e4m3 (upcast to fp32 as they are: descales must be 1), out (B,S,H,D), lse (B,H,S).
    Bounds: |O - ref| <= o_rtol * max|ref| + o_atol per head, |LSE - ref| <= lse_atol. Returns
    ``{"rows", "max_err", "max_err_lse", "tol", "ok"}`` (max over heads; "rows" = rows compared in total)."""
    B, S, H, D = q.shape
    Sk, Hk = k.shape[1], k.shape[2]
    scale = D ** -0.5 if softmax_scale is None else softmax_scale
    rows = sample_rows(S, n_rows, block_m, seed).to(q.device)
    tiles = (rows // block_m).tolist()
    res = {"rows": 0, "max_err": 0.0, "max_err_lse": 0.0, "tol": 0.0, "ok": True}
    for h in heads:
        hk = h // (H // Hk)
        kf, vf = k[batch, :, hk].float(), v[batch, :, hk].float()
        s = (q[batch, rows, h].float() @ kf.T) * scale                      # [n, Sk] fp32
        if read_list is not None:
            lists_h = read_list[batch, h].cpu()
            masks: Dict[int, torch.Tensor] = {}
            for i, m in enumerate(tiles):
                if m not in masks:
                    masks[m] = listed_key_mask(lists_h[m].tolist(), block_n, Sk, q.device)
                s[i].masked_fill_(~masks[m], float("-inf"))
        ref_lse = torch.logsumexp(s, dim=-1)
        ref_o = torch.softmax(s, dim=-1) @ vf
        err = (out[batch, rows, h].float() - ref_o).abs().max().item()
        tol = o_rtol * ref_o.abs().max().item() + o_atol
        res["max_err"] = max(res["max_err"], err)
        res["tol"] = max(res["tol"], tol)
        ok = err <= tol
        if lse is not None:
            err_l = (lse[batch, h, rows] - ref_lse).abs().max().item()
            res["max_err_lse"] = max(res["max_err_lse"], err_l)
            ok = ok and err_l <= lse_atol
        res["ok"] = res["ok"] and bool(ok)
        res["rows"] += int(rows.numel())
    res["max_err"] = float(f"{res['max_err']:.3e}")
    res["max_err_lse"] = float(f"{res['max_err_lse']:.3e}")
    res["tol"] = float(f"{res['tol']:.3e}")
    return res


# ----------------------------------------------------------------------------------------- skip vote + list writer, restated
def lists_to_bitmap(lists: torch.Tensor) -> torch.Tensor:
    """bool[..., k_tiles]: the tiles every row of ``lists`` ([..., k_tiles + 1] int32, any device) walks. Range 0 is walked even
    when L == 0 (mainloop_fwd_sm90_tma_gmma_ws.hpp:93-101); ranges are (start, end) with start >= end, both inclusive."""
    kt = lists.shape[-1] - 1
    flat = lists.reshape(-1, kt + 1).to(torch.int64)
    body = flat[:, 1:]
    if kt % 2:
        body = torch.nn.functional.pad(body, (0, 1))
    pairs = body.unflatten(-1, (-1, 2))
    live = (torch.arange(pairs.shape[1], device=lists.device)[None] < (flat[:, :1].clamp_min(2) // 2))
    starts, ends = pairs[..., 0].clamp(0, kt - 1), pairs[..., 1].clamp(0, kt - 1)
    live = live & (pairs[..., 0] >= pairs[..., 1])
    diff = torch.zeros(flat.shape[0], kt + 1, dtype=torch.int32, device=lists.device)
    one = live.to(torch.int32)
    diff.scatter_add_(1, ends, one)
    diff.scatter_add_(1, starts + 1, -one)
    return (diff.cumsum(1)[:, :kt] > 0).reshape(*lists.shape[:-1], kt)


def walk_of_row(row: Sequence[int]) -> List[int]:
    """Tiles in the order the kernel's reader visits them for one list row (mainloop...:1804-1827)."""
    n = max(int(row[0]), 2)
    seq: List[int] = []
    for i in range(1, n, 2):
        seq.extend(range(int(row[i]), int(row[i + 1]) - 1, -1))
    return seq


def written_row(read_row: Sequence[int], flags: Sequence[bool]) -> List[int]:
    """``SkipListWriter`` without a must-do list, restated (mainloop...:142-192): ``flags[i]`` is the skip vote of the i-th visited
    tile (the first visited tile is recorded as not skipped whatever it voted, :1804-1805). Returns ``[L, entries...]``."""
    out = [0]
    skipping = True
    n_ranges = max(int(read_row[0]), 2) // 2
    pos = 0
    for r in range(n_ranges):
        start, end = int(read_row[1 + 2 * r]), int(read_row[2 + 2 * r])
        skip = False
        for n in range(start, end - 1, -1):
            skip = bool(flags[pos]) and pos > 0
            pos += 1
            if skip != skipping:                      # record_transition (:163-168)
                out.append(n)
                skipping = skip
        skipping = True                               # record_range_end (:173-181)
        if not skip:
            out.append(end)
    out[0] = len(out) - 1
    return out


@torch.no_grad()
def vote_writer_check(q: torch.Tensor, k: torch.Tensor, read_list: torch.Tensor, write_list: torch.Tensor, thr: float,
                      block_m: int, block_n: int, items: Iterable, batch: int = 0, softmax_scale: Optional[float] = None,
                      margin_tol: float = 1e-3) -> Dict:
    """For every sampled ``(head, q-tile)`` of ``items``: fp32 scores of the whole q-tile (rows past seqlen_q are zero rows and
    vote, as the reference's TMA zero fill does) against all keys, the walk of the row's READ list, the skip vote of every
    walked tile — ``AND over the q-tile's rows of [(m_loc - m_prev) c <= thr]`` with the running max BEFORE the tile, c =
    softmax_scale log2 e (softmax.h:190-194) — and the write row the writer state machine produces from the votes, compared
    with the row the kernel wrote. A row that differs is "borderline" when one of its tiles has a decision margin within
    ``margin_tol`` of ``thr`` (fp32 summation order may flip it), otherwise "bad". q (B,S,H,D), k (B,Sk,Hk,D); lists
    [B,H,Qt,Kt+1]. Returns {"items", "bad", "borderline", "max_ranges", "ok"}."""
    B, S, H, D = q.shape
    Sk, Hk = k.shape[1], k.shape[2]
    kt = -(-Sk // block_n)
    c = (D ** -0.5 if softmax_scale is None else softmax_scale) * 1.4426950408889634
    res = {"items": 0, "bad": 0, "borderline": 0, "max_ranges": 0, "ok": True}
    for h, m in items:
        rows = q[batch, m * block_m: (m + 1) * block_m, h].float()
        if rows.shape[0] < block_m:
            rows = torch.nn.functional.pad(rows, (0, 0, 0, block_m - rows.shape[0]))
        s = rows @ k[batch, :, h // (H // Hk)].float().T                                  # [block_m, Sk]
        s = torch.nn.functional.pad(s, (0, kt * block_n - Sk), value=float("-inf"))       # seqlen-k mask of tile kt-1
        tile_max = s.view(block_m, kt, block_n).amax(-1)                                  # [block_m, kt]
        rd = read_list[batch, h, m].tolist()
        walk = walk_of_row(rd)
        wm = tile_max[:, torch.tensor(walk, device=q.device)]                             # row max per visited tile
        run = torch.cummax(wm, dim=1).values
        margin = ((wm[:, 1:] - run[:, :-1]) * c).amax(0)                                  # worst row of every visited tile but the first
        flags = [False] + (margin <= thr).tolist()
        want = written_row(rd, flags)
        got = write_list[batch, h, m, : want[0] + 1].tolist()
        res["items"] += 1
        res["max_ranges"] = max(res["max_ranges"], max(int(rd[0]), 2) // 2)
        if got != want:
            if bool(((margin - thr).abs() < margin_tol).any().item()):
                res["borderline"] += 1
            else:
                res["bad"] += 1
    res["ok"] = res["bad"] == 0
    return res


# ----------------------------------------------------------------------------------------- 50-step denoising workload
# Constant thresholds (log2 units) at which the list the LAST of the 50 steps reads has 21 / 42 / 57 / 77 % +- 1 % sparsity with this
# build's 256 x 64 tile on DenoiseWorkload(40 heads): bisected on all 40 heads by tools/calibrate_denoise.py, trace and result in
# profiles/r04_denoise50_calibration.json (they give 21.0 / 42.2 / 57.1 / 77.3 %; round 1's constants -5.157 / -4.22 / -3.399 / -2.462, bisected on 4 heads, gave 24.5 / 44.0 / 61.1 / 77.9 %).
DENOISE_THRESHOLDS = (("21%", -5.3438), ("42%", -4.3), ("57%", -3.6), ("77%", -2.5))
# generator="survey" (the one SURVEY.md 8(d) pins), bisected over [-20, 0) on all 40 heads: tools/calibrate_survey.py ->
# profiles/r06_denoise50_survey_calibration.json: all four targets reached within 1 % (step-49 sparsity 21.0 / 41.9 / 57.3 / 77.2 %).
SURVEY_DENOISE_THRESHOLDS = (("21%", -0.9726), ("42%", -0.7529), ("57%", -0.5478), ("77%", -0.1963))
REFERENCE_T_OVER_T0 = {"21%": 0.824, "42%": 0.601, "57%": 0.443, "77%": 0.235}     # /root/reference/README.md:81-87


class DenoiseWorkload:
    """BASELINE.json configs[2]: synthetic, slowly varying, STRUCTURED q/k/v of a 50-step denoising loop at the Wan2.1 video shape
    (iid randn gives ~0 % sparsity at any negative threshold). S = 21 frames x 3600 tokens; step t: x_t = sqrt(1 - s_t^2) x0 + s_t n_t,
    s_t linear 0.5 -> 0.05, noise seed 10^6 + t. Two generators of x0:

    ``generator="anchored"`` (default; the committed 50-step runs since round 1, tools/denoise_bench.py --alpha 6 --sink-gain 0.5): per
    head the frame centroids follow an AR(1) walk (scores decay smoothly with frame distance), alpha = 6, and the last `sink` tokens are
    global anchor keys (QK-Skip walks keys in descending order and can only drop tiles met after a row's dominant keys).

    ``generator="survey"``: the generator SURVEY.md 8(d) pins - per head h (seed 1234 + h) centroids = normalised randn smoothed over the
    frames with [0.25, 0.5, 0.25], q0 = alpha u_f(i) + randn, k0 likewise, v0 = randn, alpha = 4, no anchor keys. Its structure is
    weak (a same-frame score is 0.5 nat above a cross-frame one against a per-tile maximum of 4 sigma of noise): at the thresholds of the
    anchored runs it skips almost nothing, which is why round 1 left it; profiles/r04_denoise50_calibration.json holds both side by side."""
    FRAMES, PER, D = 21, 3600, 128

    def __init__(self, heads: int, device, steps: int = 50, alpha: Optional[float] = None, rho: float = 0.85, sink: int = 640,
                 sink_gain: float = 0.5, seed: int = 1234, generator: str = "anchored"):
        self.steps, self.device, self.generator = steps, device, generator
        S = self.S = self.FRAMES * self.PER
        frame = torch.arange(S, device=device) // self.PER
        if generator == "survey":
            alpha = 4.0 if alpha is None else alpha
            q0 = torch.empty(S, heads, self.D, device=device)
            k0, v0 = torch.empty_like(q0), torch.empty_like(q0)
            for h in range(heads):
                g = torch.Generator(device=device).manual_seed(seed + h)
                u = torch.randn(self.FRAMES, self.D, device=device, generator=g)
                u = u / u.norm(dim=-1, keepdim=True)
                pad = torch.cat([u[:1], u, u[-1:]])                                   # edge frames: replicate
                u = 0.25 * pad[:-2] + 0.5 * pad[1:-1] + 0.25 * pad[2:]
                q0[:, h] = alpha * u[frame] + torch.randn(S, self.D, device=device, generator=g)
                k0[:, h] = alpha * u[frame] + torch.randn(S, self.D, device=device, generator=g)
                v0[:, h] = torch.randn(S, self.D, device=device, generator=g)
            self.base = (q0[None], k0[None], v0[None])
            return
        if generator != "anchored":
            raise ValueError("generator must be 'anchored' or 'survey'")
        alpha = 6.0 if alpha is None else alpha
        g = torch.Generator(device=device).manual_seed(seed)
        z = torch.randn(self.FRAMES, heads, self.D, device=device, generator=g)
        u = torch.empty_like(z)
        u[0] = z[0]
        for f in range(1, self.FRAMES):
            u[f] = rho * u[f - 1] + (1 - rho ** 2) ** 0.5 * z[f]
        u = u / u.norm(dim=-1, keepdim=True)
        cen = u[frame]
        q0 = alpha * cen + torch.randn(S, heads, self.D, device=device, generator=g)
        k0 = alpha * cen + torch.randn(S, heads, self.D, device=device, generator=g)
        anchor = u.mean(0)
        anchor = anchor / anchor.norm(dim=-1, keepdim=True)
        q0 = q0 + alpha * sink_gain * anchor
        k0[S - sink:] = k0[S - sink:] + alpha * (1 + sink_gain) * anchor
        v0 = torch.randn(S, heads, self.D, device=device, generator=g)
        self.base = (q0[None], k0[None], v0[None])

    def qkv(self, t: int):
        s = 0.5 + (0.05 - 0.5) * t / max(1, self.steps - 1)
        g = torch.Generator(device=self.device).manual_seed(10 ** 6 + t)
        return [((1 - s * s) ** 0.5 * x + s * torch.randn(x.shape, device=self.device, generator=g)).to(torch.bfloat16)
                for x in self.base]
