import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import nextgenmap_amd as N
from pairgen import make_pairs
AFF = int(os.environ.get('AFFINE','0'))
kw = dict(personality=1, gap_read=33, gap_ref=33, gap_extend=3) if AFF else {}
for (q,c,rl) in [(102,20,100),(152,27,150),(252,42,250)]:
    n = 1<<20
    br, bq = make_pairs(8192, q, c, seed=1, read_len=rl, mix=(0.7,0.3,0.0))
    idx = np.random.default_rng(0).integers(0,8192,n)
    ref = torch.from_numpy(br[idx]).cuda(); qry = torch.from_numpy(bq[idx]).cuda()
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    eng = N.Engine(q,c,max_batch=n,**kw); eng.set_profiling(True)
    st = torch.cuda.current_stream().cuda_stream
    for mode in (0,1):
        for it in range(3):
            eng.score_device(mode,n,ref,qry,out,st); torch.cuda.synchronize()
            ms = eng.last_kernel_ms()
        cells = n*rl*c
        print("q=%d c=%d mode=%d n=%d pack %.3f ms dp %.3f ms -> %.2f Mpairs/s dp-only, %.1f Gcells/s; pack+dp %.2f Mpairs/s" % (q,c,mode,n,ms[0],ms[1], n/ms[1]/1e3, cells/ms[1]/1e6, n/(ms[0]+ms[1])/1e3))
    rs = eng.align_run_stride()
    na = 1<<18
    rec = torch.empty((na,8),dtype=torch.int32,device="cuda"); runs = torch.empty((na,rs),dtype=torch.int16,device="cuda")
    for it in range(2):
        eng.align_device(0,na,ref,qry,rec,runs,rs,st); torch.cuda.synchronize(); ms = eng.last_kernel_ms()
    print("   align n=%d pack %.3f dp %.3f tb %.3f ms -> %.2f Mpairs/s" % (na, ms[0],ms[1],ms[2], na/sum(ms)/1e3))
    eng.close()
