"""Bring-up script (run on the GPU box): hardware probes + first parity numbers vs the CPU oracle."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def probes():
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libprobes.so"))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    addr = (torch.arange(64, dtype=torch.int32) * 8).cuda()
    out = torch.zeros(256, dtype=torch.int16, device="cuda")
    lib.probe_tr16(ctypes.c_void_p(addr.data_ptr()), ctypes.c_void_p(out.data_ptr()), st)
    torch.cuda.synchronize()
    o = out.cpu().view(64, 4)
    print("tr16 lane-linear (elem index = lane*4+e): lanes 0..17")
    for l in list(range(18)) + [32, 33]:
        print(l, o[l].tolist())
    # expected under the assumed semantics: out[i][j] = in[4j + i/4][i%4] within each 16-lane group
    exp = torch.zeros(64, 4, dtype=torch.int16)
    for l in range(64):
        g, i = l // 16, l % 16
        for j in range(4):
            src_lane = 16 * g + 4 * j + i // 4
            exp[l, j] = src_lane * 4 + i % 4
    print("tr16 semantics as assumed:", torch.equal(o, exp))
    A = torch.randint(-4, 5, (32, 16)).float(); B = torch.randint(-4, 5, (16, 32)).float()
    C = torch.zeros(32, 32, device="cuda")
    Ad, Bd = A.cuda(), B.cuda()
    lib.probe_mfma(ctypes.c_void_p(Ad.data_ptr()), ctypes.c_void_p(Bd.data_ptr()), ctypes.c_void_p(C.data_ptr()), st)
    torch.cuda.synchronize()
    print("mfma layout as assumed:", torch.equal(C.cpu(), A @ B))

def parity():
    import liteattention_amd as L
    from oracle import oracle as orc
    torch.manual_seed(0)
    for (B, S, H) in [(1, 256, 1), (1, 1000, 2), (2, 333, 3)]:
        q, k, v = [torch.randn(B, S, H, 128).bfloat16() for _ in range(3)]
        o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=128, block_n=64)
        o, lse = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
        torch.cuda.synchronize()
        print(f"dense B{B} S{S} H{H}: max|o-oracle| = {(o.float().cpu()-o_ref).abs().max():.3e}  "
              f"max|lse-oracle| = {(lse.cpu()-lse_ref).abs().max():.3e}")
    # skip lists
    B, S, H = 1, 1000, 2
    q, k, v = [torch.randn(B, S, H, 128).bfloat16() for _ in range(3)]
    for thr in [float("inf"), float("-inf"), -1.0, 0.0]:
        att = L.LiteAttention(max_batch_size=1)
        att.threshold = thr
        o = att(q.cuda(), k.cuda(), v.cuda())
        torch.cuda.synchronize()
        sl = orc.init_skip_list_ref(1, 8, 16, H)
        orc.qkskip_fwd(q, k, v, block_m=128, block_n=64, read_list=sl[0], write_list=sl[1], thr=thr)
        same = torch.equal(att._skip_list[1].cpu(), sl[1])
        print(f"thr={thr}: write list == oracle: {same}; row0 = {att._skip_list[1,0,0,0,:8].tolist()}")
        if not same:
            print("  oracle row0:", sl[1, 0, 0, 0, :8].tolist())

def quick_bench():
    import liteattention_amd as L
    for S, H in [(8192, 8), (32768, 40)]:
        q, k, v = [torch.randn(1, S, H, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3)]
        for _ in range(2): L.flash_attn_func(q, k, v)
        torch.cuda.synchronize()
        t = time.time(); n = 3
        for _ in range(n): L.flash_attn_func(q, k, v)
        torch.cuda.synchronize()
        dt = (time.time() - t) / n
        print(f"dense S={S} H={H}: {dt*1e3:.2f} ms  {4*H*S*S*128/dt/1e12:.1f} TFLOP/s")

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    probes(); parity(); quick_bench()
