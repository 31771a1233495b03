#!/usr/bin/env python3
"""What a user of the reference does to an nn.Module besides forward / backward: each probe prints ok / the exception."""
import copy, io, pickle, sys, traceback
import torch, torch.nn as nn
sys.path.insert(0, ".")
from music_amd.model import wavenet
from music_amd.model1 import wavenet_autoencoder

CFG = dict(filter_width=2, dilations=[1, 2, 4, 8], dilation_channels=32, residual_channels=32, skip_channels=64, quantization_channels=256, use_bias=False)


def probe(name, fn):
    try:
        r = fn()
        print("%-44s ok %s" % (name, "" if r is None else r))
    except Exception as e:
        print("%-44s %s: %s" % (name, type(e).__name__, str(e).split("\n")[0][:150]))


def x(B=2, T=300, dev="cuda"):
    g = torch.Generator().manual_seed(1)
    c = torch.randint(0, 256, (B, T), generator=g)
    return torch.nn.functional.one_hot(c, 256).permute(0, 2, 1).float().contiguous().to(dev)


torch.manual_seed(0)
net = wavenet(**CFG).cuda()
rf = net.receptive_field
ref = net(x()).detach().clone()
tgt = torch.randint(0, 256, (ref.shape[0],), device="cuda")

def p_eval():
    net.eval()
    with torch.no_grad():
        o = net(x())
    net.train()
    return float((o - ref).abs().max())
probe("eval + no_grad forward", p_eval)

def p_dp():
    dp = nn.DataParallel(net, device_ids=[0])
    o = dp(x())
    loss = nn.CrossEntropyLoss()(o, tgt)
    loss.backward()
    return float((o.detach() - ref).abs().max())
probe("nn.DataParallel(net, [0]) fwd + bwd", p_dp)

def p_deepcopy():
    n2 = copy.deepcopy(net)
    o = n2(x())
    nn.CrossEntropyLoss()(o, tgt).backward()
    return float((o.detach() - ref).abs().max())
probe("copy.deepcopy(net) fwd + bwd", p_deepcopy)

def p_pickle():
    buf = io.BytesIO()
    torch.save(net, buf)
    buf.seek(0)
    n2 = torch.load(buf, weights_only=False)
    o = n2(x())
    return float((o.detach() - ref).abs().max())
probe("torch.save(net) / torch.load whole module", p_pickle)

def p_to():
    n2 = wavenet(**CFG)
    n2.load_state_dict(net.state_dict())
    n2 = n2.to("cuda:0")
    return float((n2(x()).detach() - ref).abs().max())
probe("load_state_dict + .to('cuda:0')", p_to)

def p_cpu_roundtrip():
    n2 = copy.deepcopy(net).cpu().cuda()
    o = n2(x())
    nn.CrossEntropyLoss()(o, tgt).backward()
    return float((o.detach() - ref).abs().max())
probe(".cpu().cuda() round trip fwd + bwd", p_cpu_roundtrip)

probe("forward on CPU tensors (must raise)", lambda: copy.deepcopy(net).cpu()(x(dev="cpu")))
probe("net.half() forward (must raise)", lambda: copy.deepcopy(net).half()(x().half()))
probe("net.double() forward (must raise)", lambda: copy.deepcopy(net).double()(x().double()))
probe("input shorter than the receptive field", lambda: net(x(T=rf - 1)).shape)
probe("input exactly the receptive field", lambda: tuple(net(x(T=rf)).shape))
probe("non-contiguous input (a permuted view)", lambda: float((net(x().permute(0, 2, 1).contiguous().permute(0, 2, 1)).detach() - ref).abs().max()))
probe("requires_grad input: d loss / d input", lambda: (lambda xi: (nn.CrossEntropyLoss()(net(xi), tgt).backward(), float(xi.grad.abs().sum()))[1])(x().requires_grad_(True)))

def p_zero_grad_none():
    o = net(x()); nn.CrossEntropyLoss()(o, tgt).backward()
    net.zero_grad(set_to_none=True)
    o = net(x()); nn.CrossEntropyLoss()(o, tgt).backward()
    return float(sum(p.grad.abs().sum() for p in net.parameters()))
probe("zero_grad(set_to_none=True) between steps", p_zero_grad_none)

def p_frozen():
    n2 = copy.deepcopy(net)
    n2.causal_layer.weight.requires_grad_(False)
    o = n2(x()); nn.CrossEntropyLoss()(o, tgt).backward()
    return n2.causal_layer.weight.grad is None
probe("a frozen parameter keeps grad None", p_frozen)

def p_clip():
    o = net(x()); nn.CrossEntropyLoss()(o, tgt).backward()
    return float(torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0))
probe("clip_grad_norm_", p_clip)

def p_amp():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        o = net(x())
        loss = nn.CrossEntropyLoss()(o, tgt)
    loss.backward()
    return str(o.dtype)
probe("torch.autocast region", p_amp)

def p_sd_keys():
    return len(net.state_dict()), list(net.state_dict())[:2]
probe("state_dict keys", p_sd_keys)

torch.manual_seed(0)
ae = wavenet_autoencoder(filter_width=2, quantization_channel=256, dilations=[1, 2, 4, 8], en_residual_channel=32, en_dilation_channel=32,
                         en_bottleneck_width=16, en_pool_kernel_size=100, de_residual_channel=32, de_dilation_channel=32, de_skip_channel=64,
                         use_bias=False).cuda()

def ae_fwd(n):
    torch.manual_seed(5)
    return n(x(T=400))
aref = ae_fwd(ae).detach().clone()
atgt = torch.randint(0, 256, (aref.shape[0],), device="cuda")
probe("autoencoder fwd + bwd", lambda: (nn.CrossEntropyLoss()(ae_fwd(ae), atgt).backward(), tuple(aref.shape))[1])
probe("autoencoder deepcopy fwd", lambda: float((ae_fwd(copy.deepcopy(ae)).detach() - aref).abs().max()))
def p_ae_pickle():
    buf = io.BytesIO(); torch.save(ae, buf); buf.seek(0)
    return float((ae_fwd(torch.load(buf, weights_only=False)).detach() - aref).abs().max())
probe("autoencoder torch.save / load whole module", p_ae_pickle)
probe("autoencoder DataParallel([0]) fwd + bwd", lambda: (lambda o: (nn.CrossEntropyLoss()(o, atgt).backward(), float((o.detach() - aref).abs().max()))[1])((torch.manual_seed(5), nn.DataParallel(ae, device_ids=[0])(x(T=400)))[1]))
probe("autoencoder eval + no_grad", lambda: (ae.eval(), torch.no_grad().__enter__(), float((ae_fwd(ae) - aref).abs().max()), torch.set_grad_enabled(True), ae.train())[2])
