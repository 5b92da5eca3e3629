import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion324_amd import ops
torch.manual_seed(0)
M, N, K = 128, 128, 64
# A[m][k] = m + k/100 ; W = one-hot rows: W[n][k] = (k == n % K)  -> out[m][n] = A[m][n % K]
a = (torch.arange(M).float()[:, None] + torch.arange(K).float()[None, :] / 100)
w = torch.zeros(N, K); w[torch.arange(N), torch.arange(N) % K] = 1
out = torch.full((M, N), -1.0, device="cuda")
ops.gemm(a.cuda(), w.cuda(), out)
o = out.cpu()
ref = a @ w.T
print("max err", (o - ref).abs().max().item())
print("out[0:3, 0:10]", o[0:3, 0:10])
print("ref[0:3, 0:10]", ref[0:3, 0:10])
# decode: value = m' + k'/100
mm = o.floor(); kk = ((o - mm) * 100).round()
print("decoded m'[0:40:1, 0]", mm[0:40, 0].tolist())
print("decoded k'[0, 0:64]", kk[0, 0:64].tolist())
print("decoded k'[1, 0:64]", kk[1, 0:64].tolist())
print("decoded k'[2, 0:16]", kk[2, 0:16].tolist())
# now the other way: A one-hot, W coded -> which W element lands where
a2 = torch.zeros(M, K); a2[torch.arange(M), torch.arange(M) % K] = 1
w2 = (torch.arange(N).float()[:, None] + torch.arange(K).float()[None, :] / 100)
out2 = torch.full((M, N), -1.0, device="cuda")
ops.gemm(a2.cuda(), w2.cuda(), out2)
o2 = out2.cpu(); ref2 = a2 @ w2.T
print("max err2", (o2 - ref2).abs().max().item())
nn_ = o2.floor(); kk2 = ((o2 - nn_) * 100).round()
print("decoded n'[0, 0:40]", nn_[0, 0:40].tolist())
print("decoded k2'[0:64, 0]", kk2[0:64, 0].tolist())
