"""Phase stamps of k_subm_halo64 (variant built with -DHL_PHASE_TIMING: tools/build_file_variant.sh hl_phase subm_halo -DHL_PHASE_TIMING;
run with U3D_LIB_PATH=uni3detr_amd/_variants/hl_phase.so): shader-clock deltas between the marks of workgroups 7, 263, 519, ...
(one per scheduling round of 256 CUs x 2)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uni3detr_amd import native as nv  # noqa: E402
from uni3detr_amd import sparse as sp  # noqa: E402


def main():
    torch.manual_seed(0)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sparse_bench
    lvl = [g.level for _, ci, co, g in sparse_bench.build_jobs(8, torch.device("cuda:0")) if ci == 64 and co == 64 and g.level is not None][0]
    halo = lvl.halo()
    tc = halo.tile_cnt.float()
    print(f"rows {lvl.n} tiles {halo.tiles} distinct/tile mean {tc.mean().item():.0f} max {tc.max().item():.0f}")
    for _ in range(3):
        nv.SubmHalo(lvl.subm_tables()[0], lvl.n_dev, lvl.n)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 64)()
    assert nv.lib().u3d_debug_halo_build_times(buf) == 0
    names = ["loads+clear", "mark bits", "popcount scan", "emit rows", "slots"]
    for r in range(8):
        m = [buf[r * 8 + i] for i in range(6)]
        if m[5]:
            print(f"build wg {r * 256 + 7}: start +{m[0] - buf[0]:>8d} | " + " | ".join(f"{nm} {m[i + 1] - m[i]:>6d}" for i, nm in enumerate(names)) + f" | total {m[5] - m[0]}")
    x = torch.randn(lvl.n, 64, device="cuda").bfloat16()
    dyv = torch.randn(lvl.n, 64, device="cuda").bfloat16()
    for _ in range(3):
        nv.subm_halo_wgrad(x, dyv, halo)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 64)()
    if nv.lib().u3d_debug_halo_wgrad_times(buf) == 0:
        m = list(buf)
        print(f"wgrad wg 7: prologue {m[1] - m[0]}")
        it = 0
        while 5 + it * 5 < 64 and m[5 + it * 5]:
            b = 1 + it * 5
            print(f"  tile {it}: wait+barrier {m[b + 1] - m[b]:>6d} | issue next {m[b + 2] - m[b + 1]:>6d} | dy fragments {m[b + 3] - m[b + 2]:>6d} | offsets {m[b + 4] - m[b + 3]:>6d}")
            it += 1
        print(f"  tail: wait {0} partial store {m[2 + it * 5] - m[1 + it * 5]}")
    w = nv.subm_halo_wpack((torch.randn(27, 64, 64, device="cuda") * 0.1).bfloat16())
    for stats in (False, True):
        for _ in range(3):
            nv.subm_halo_conv(x, w, halo, want_stats=stats)
        torch.cuda.synchronize()
        buf = (C.c_uint64 * 64)()
        assert nv.lib().u3d_debug_halo_times(buf) == 0
        names = ["start->staged", "barrier", "offset loop", "reduce-scatter", "epilogue"]
        for r in range(8):
            m = [buf[r * 8 + i] for i in range(6)]
            if m[5] == 0:
                continue
            print(f"stats={stats} wg {r * 256 + 7}: start +{m[0] - buf[0]:>8d} | " + " | ".join(f"{nm} {m[i + 1] - m[i]:>6d}" for i, nm in enumerate(names))
                  + f" | total {m[5] - m[0]}")




def emptiness():
    """Share of (tile, offset, 16-row block) triples without any neighbour, and of whole (tile, offset) pairs (MFMA work that could be skipped)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sparse_bench
    lvl = [g.level for _, ci, co, g in sparse_bench.build_jobs(8, torch.device("cuda:0")) if ci == 64 and co == 64 and g.level is not None][0]
    halo = lvl.halo()
    T = halo.tiles
    loc = (halo.loc.view(T, 27, 16, 8).long() & 0xffff)          # [tile][k][r16][mt]
    blk = (loc != 0).any(2)                                         # [tile][k][mt]
    print(f"non-empty 16-row blocks: {blk.float().mean().item():.3f}; non-empty (tile, offset) pairs: {blk.any(2).float().mean().item():.3f}; "
          f"non-empty 32-row pairs of blocks: {(blk.view(T, 27, 4, 2).any(3)).float().mean().item():.3f}; present pairs {(loc != 0).float().mean().item():.3f}")
    per_k = blk.float().mean((0, 2))
    print("per offset:", " ".join(f"{v:.2f}" for v in per_k.tolist()))


if len(sys.argv) > 1 and sys.argv[1] == "empty":
    emptiness()
elif __name__ == "__main__":
    main()
