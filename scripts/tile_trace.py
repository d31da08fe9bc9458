"""Timeline of one wave of one block of the split conv (probe build, SGDFR_LIB=.../libsgdfr_hip_probe.so): s_memtime stamps at
the phase boundaries of every tile the block runs.  Prints the average shader clocks per phase over the steady-state tiles.
    events: 1 tile start | 2 descriptors done | 8 prologue issued, before the wait+barrier | 3 K loop starts | 16+u sub-stage u |
            4 K loop done | 5 element loop done | 6 ToRGB reduce done | 7 end-of-tile barrier passed"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import functional as F_, _native as N
B = int(os.environ.get('B', 64))
SUB = int(os.environ.get('SUB', 0))
FINE = int(os.environ.get('FINE', 0))     # stamps inside a sub-stage: 32 DMA issued | 33 MFMAs issued | 34 vmcnt wait done | 35 barrier passed
LAYERS = [(512, 512, 32, 0, 1, 1, 0), (256, 256, 64, 0, 1, 1, 0), (128, 128, 128, 0, 1, 1, 0), (64, 64, 256, 0, 1, 0, 0),
          (512, 512, 16, 1, 0, 0, 1), (512, 256, 32, 1, 0, 0, 1), (256, 128, 64, 1, 0, 0, 1), (128, 64, 128, 1, 0, 0, 1)]
NAMES = {32: 'dma-issue', 33: 'mfma', 34: 'vmwait', 35: 'barrier', 1: 'start', 2: 'descr', 8: 'prologue-issue', 3: 'prologue-wait', 4: 'kloop', 5: 'elements', 6: 'rgb', 7: 'endsync'}
lib = N.load()
lib.sgdfr_split_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
for cin, cout, h, up, rgb, exs, wy in [LAYERS[int(i)] for i in os.environ.get('LAYERS', '0,1,2,3,4,5,6,7').split(',')]:
    w = torch.randn(1, cout, cin, 3, 3, device='cuda'); x = torch.randn(B, cin, h, h, device='cuda')
    s = torch.randn(B, cin, device='cuda'); d = torch.rand(B, cout, device='cuda') + 0.5
    nz = torch.randn(1, 1, h, h, device='cuda'); nw = torch.full((1,), 0.1, device='cuda'); bias = torch.randn(cout, device='cuda')
    wsp = F_.prepack_split(w, 'fp16x3'); xs = F_.to_split(x, s, 'fp16x3'); del x
    if up:
        ps = ((h + 1) * (h + 1) + 31) // 32 * 32
        buf = torch.empty((B, cout, 4, ps), device='cuda')
        fn = lambda: F_.modconv_split(xs, wsp, None, d, cout, arith='fp16x3', mode=N.MODE_UP3, x_split=(B, cin, h, h), batch=B, out=buf, plane_stride=ps)
    else:
        rw = torch.randn(3, cout, device='cuda'); rs = torch.randn(B, cout, device='cuda'); sn = torch.randn(B, cout, device='cuda')
        fn = lambda: F_.modconv_split(xs, wsp, None, d, cout, nz, nw, bias, True, arith='fp16x3', x_split=(B, cin, h, h), batch=B,
                                      rgb=(rw, rs) if rgb else None, s_next=sn if exs else None, want_y=bool(wy))
    for blk, wave in [tuple(int(v) for v in bw.split(':')) for bw in os.environ.get('BW', '0:0,0:7,133:3').split(',')]:
        os.environ['SGDFR_SPLIT_DBG'] = str(64 | (128 if SUB else 0) | (32 if FINE else 0) | (blk << 8) | (wave << 16))
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        buf64 = (ctypes.c_ulonglong * 4096)()
        assert lib.sgdfr_split_trace_read(buf64, 4096) == 0
        n = min(int(buf64[0]), 4094)
        ev = [(int(buf64[1 + i]) >> 48, int(buf64[1 + i]) & ((1 << 48) - 1)) for i in range(n)]
        # split into tiles at event 1
        tiles, cur = [], []
        for e in ev:
            if e[0] == 1 and cur: tiles.append(cur); cur = []
            cur.append(e)
        if cur: tiles.append(cur)
        steady = tiles[1:-1] if len(tiles) > 3 else tiles
        acc, cnt = {}, 0
        for i, t in enumerate(steady):
            nxt = None
            for j in range(1, len(t)):
                acc[t[j][0]] = acc.get(t[j][0], 0) + (t[j][1] - t[j - 1][1])
            cnt += 1
        total = sum(acc.values()) / max(cnt, 1)
        line = ' '.join('%s=%d' % (NAMES.get(k, 'ss%d' % (k - 16)), v / max(cnt, 1)) for k, v in sorted(acc.items(), key=lambda kv: (kv[0] >= 16, kv[0])))
        span = (ev[-1][1] - ev[0][1]) if ev else 0
        print('%s%d->%d@%d blk %d wave %d: %d tiles, launch %.0f us, traced span %d clk | per tile %d clk: %s' % (
            'up' if up else 'pl', cin, cout, h, blk, wave, len(tiles), e0.elapsed_time(e1) * 1e3, span, total, line), flush=True)
os.environ['SGDFR_SPLIT_DBG'] = '0'
