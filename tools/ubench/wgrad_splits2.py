import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tools.kbench as kb, torch
kb._lib.require_gpu(); torch.zeros(1, device="cuda")
E=512; M=128*1024
for (m,n) in ((E,4*E),(E,3*E),(E,E)):
    for flags, splits in ((0,16),(0,21),(0,48),(4,4),(4,5),(4,8),(4,16),(32|16,8),(32|16,10),(32|16,16)):
        kb.GFLAGS = flags
        try:
            kb.gemm_case("wgrad %dx%d flags=%d" % (m,n,flags), 1, 0, m, n, M, splitk=splits, out_fp32=True)
        except Exception as e:
            print("fail", flags, splits, e)
