"""Which GELU does the library GEMM's epilogue (torch._addmm_activation, bf16) apply — erf or tanh?  Identity weights so the
pre-activation is the input itself; compare against both forms on a dense grid."""
import torch
dev = "cuda"
n = 256
x = torch.linspace(-6, 6, 4096 * n, device=dev).view(4096, n).bfloat16()
W = torch.eye(n, device=dev).bfloat16()
b = torch.zeros(n, device=dev).bfloat16()
y = torch._addmm_activation(b, x, W.t(), use_gelu=True).float()
xf = x.float()
erf = torch.nn.functional.gelu(xf)
tanh = torch.nn.functional.gelu(xf, approximate="tanh")
print("vs erf  gelu: max", float((y - erf.bfloat16().float()).abs().max()), "mean", float((y - erf).abs().mean()))
print("vs tanh gelu: max", float((y - tanh.bfloat16().float()).abs().max()), "mean", float((y - tanh).abs().mean()))
print("erf vs tanh (fp32): max", float((erf - tanh).abs().max()))
