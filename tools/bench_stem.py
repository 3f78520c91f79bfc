import torch, sys
sys.path.insert(0, "/root/repo")
from gkgnet_amd import fused, layers
layers.norm_cfg["type"] = "BN"
conv = torch.nn.Conv2d(3, 40, 3, stride=2, padding=1).cuda()
bn = layers.build_norm(40).cuda().eval()
x = torch.randn(32, 3, 576, 576, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with torch.no_grad():
    print("own conv fp32 out", t(lambda: fused.stem_conv(conv, x)))
    print("own conv+bn+gelu bf16 out", t(lambda: fused.stem_conv_bn_act_eval(conv, bn, torch.nn.GELU(), x, True)))
    print("own conv+bn+gelu fp32 out", t(lambda: fused.stem_conv_bn_act_eval(conv, bn, torch.nn.GELU(), x, False)))
    print("miopen fp32", t(lambda: conv(x)))
    xc = x.contiguous(memory_format=torch.channels_last)
    print("miopen fp32 channels_last", t(lambda: conv(xc)))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        print("miopen bf16 autocast", t(lambda: conv(x)))


def bench_backward():
    """weight gradient of the first stem convolution through the library (what _StemConv.backward calls), channels-last operands"""
    import torch
    x = torch.randn(32, 3, 576, 576, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(40, 3, 3, 3, device="cuda")
    g = torch.randn(32, 40, 288, 288, device="cuda").contiguous(memory_format=torch.channels_last)

    def f():
        return torch.ops.aten.convolution_backward(g, x, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); e1.synchronize()
    print("library weight gradient of the first stem convolution (channels-last operands)", e0.elapsed_time(e1) * 100, "us")


if __name__ == "__main__":
    bench_backward()
