#!/usr/bin/env python3
"""More of what a user does to the module: accumulation, retain_graph, autograd.grad, hooks, re-initialisation, streams."""
import sys
import torch, torch.nn as nn
sys.path.insert(0, ".")
from music_amd.model import wavenet
from music_amd.model1 import wavenet_autoencoder

CFG = dict(filter_width=2, dilations=[1, 2, 4, 8, 32], dilation_channels=32, residual_channels=32, skip_channels=64, quantization_channels=256, use_bias=True)


def probe(name, fn):
    try:
        r = fn()
        print("%-52s ok %s" % (name, "" if r is None else r))
    except Exception as e:
        print("%-52s %s: %s" % (name, type(e).__name__, str(e).split("\n")[0][:160]))


g = torch.Generator().manual_seed(1)
torch.manual_seed(0)
net = wavenet(**CFG)
with torch.no_grad():
    for p in net.parameters():
        p.mul_(3.0)
opt_before = torch.optim.SGD(net.parameters(), lr=0.1)       # created BEFORE the module moves / its engine exists
net = net.cuda()
T = net.receptive_field + 300
x = (torch.randn(2, 256, T, generator=g) * 0.5).cuda()
W = T - net.receptive_field + 1
tgt = torch.randint(0, 256, (2 * W,), generator=g).cuda()
ce = nn.CrossEntropyLoss()


def grads():
    return torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()


net.zero_grad(); ce(net(x), tgt).backward(); g1 = grads()

def p_accum():
    net.zero_grad()
    ce(net(x), tgt).backward(); ce(net(x), tgt).backward()
    return float((grads() - 2 * g1).abs().max() / g1.abs().max())
probe("two backward passes accumulate into .grad", p_accum)

def p_retain():
    net.zero_grad()
    loss = ce(net(x), tgt)
    loss.backward(retain_graph=True); loss.backward()
    return float((grads() - 2 * g1).abs().max() / g1.abs().max())
probe("backward(retain_graph=True) then backward again", p_retain)

def p_autograd_grad():
    loss = ce(net(x), tgt)
    gs = torch.autograd.grad(loss, list(net.parameters()))
    return float((torch.cat([a.reshape(-1) for a in gs]) - g1).abs().max() / g1.abs().max())
probe("torch.autograd.grad(loss, parameters)", p_autograd_grad)

def p_scaled():
    net.zero_grad()
    (ce(net(x), tgt) * 0.5).backward()
    return float((grads() - 0.5 * g1).abs().max() / g1.abs().max())
probe("a scaled loss", p_scaled)

def p_param_hook():
    seen = []
    h = net.causal_layer.weight.register_hook(lambda gr: seen.append(float(gr.abs().sum())) or gr * 2)
    net.zero_grad(); ce(net(x), tgt).backward(); h.remove()
    ratio = float(net.causal_layer.weight.grad.abs().sum()) / seen[0]
    return "hook saw the gradient, doubled it: ratio %.3f" % ratio
probe("parameter.register_hook", p_param_hook)

def p_fwd_hook():
    seen = []
    h = net.register_forward_hook(lambda m, i, o: seen.append(tuple(o.shape)))
    net(x); h.remove()
    return seen
probe("module.register_forward_hook", p_fwd_hook)

def p_opt_before():
    net.zero_grad(); ce(net(x), tgt).backward()
    w0 = net.post_process_2.weight.detach().clone()
    opt_before.step()
    moved = float((net.post_process_2.weight.detach() - w0).abs().max())
    out_changed = float((net(x).detach() - ref0).abs().max())
    return "weight moved %.2e, output moved %.2e" % (moved, out_changed)
ref0 = net(x).detach().clone()
probe("optimizer built before .cuda() / the first forward", p_opt_before)

def p_reinit():
    with torch.no_grad():
        net.apply(lambda m: nn.init.normal_(m.weight, std=0.05) if isinstance(m, nn.Conv1d) else None)
    o1 = net(x).detach().clone()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    n2 = wavenet(**CFG); n2.load_state_dict(sd); n2 = n2.cuda()
    return float((n2(x).detach() - o1).abs().max())
probe("net.apply(init) after the engine exists", p_reinit)

def p_load_after():
    sd = {k: torch.randn_like(v) * 0.05 for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    n2 = wavenet(**CFG); n2.load_state_dict(sd); n2 = n2.cuda()
    return float((n2(x).detach() - net(x).detach()).abs().max())
probe("load_state_dict after the engine exists", p_load_after)

def p_stream():
    s = torch.cuda.Stream()
    o0 = net(x).detach().clone()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        o = net(x)
        l = ce(o, tgt); l.backward()
    torch.cuda.current_stream().wait_stream(s)
    return float((o.detach() - o0).abs().max())
probe("forward + backward on a side stream", p_stream)

def p_vec():
    v = nn.utils.parameters_to_vector(net.parameters())
    nn.utils.vector_to_parameters(v * 1.0, net.parameters())
    return float((net(x).detach() - net(x).detach()).abs().max())
probe("parameters_to_vector / vector_to_parameters", p_vec)

def p_float_target():
    return float(ce(net(x), torch.softmax(torch.randn(2 * W, 256, device="cuda"), 1)))
probe("CrossEntropyLoss with probability targets", p_float_target)

def p_nll():
    o = net(x)
    l = nn.functional.nll_loss(torch.log(o + 1e-12), tgt); l.backward()
    return float(l)
probe("another loss on the output (nll of log)", p_nll)

def p_requires_grad_false_all():
    for p in net.parameters(): p.requires_grad_(False)
    o = net(x)
    r = o.requires_grad
    for p in net.parameters(): p.requires_grad_(True)
    return "output.requires_grad = %s" % r
probe("all parameters frozen", p_requires_grad_false_all)

def p_train_flag():
    return "training=%s after eval(): %s" % (net.training, net.eval().training) + " / " + str(net.train().training)
probe("train() / eval() flags", p_train_flag)

def p_named():
    return len(list(net.named_modules())), len(list(net.children()))
probe("named_modules / children", p_named)

def p_repr():
    return len(repr(net))
probe("repr(net)", p_repr)
