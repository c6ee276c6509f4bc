"""`hifigan.Generator` drop-in (reference hifigan/models.py:112-173, hifigan/__init__.py:4-7): mel -> waveform.

Same constructor (`Generator(h)` with the hifigan/config.json fields), `forward(x[B, 80, T]) -> [B, 1, T * prod(rates)]`,
`remove_weight_norm()` and state-dict layout (`conv_pre / ups.i / resblocks.j.convs{1,2}.m / conv_post` with
`weight_g`, `weight_v`, `bias`; `weight` after `remove_weight_norm`), so the released generator checkpoints
(utils.py:251-262) load unchanged.  Inference only (the reference never trains the vocoder).

Every convolution runs on the implicit-GEMM engine of libstyler_hip.so (`styler_conv_gemm_pad`), time-major [L, C]:
  * Conv1d(k, dilation d): the d phase views x[r::d] of the sequence are ordinary k-tap convs (row stride d*C);
  * an 11-tap conv is two 6-tap calls: taps 0..5 (pad 5), then taps 5..10 with the shared tap zeroed (pad 0)
    accumulating through the residual input of the epilogue;
  * ConvTranspose1d(k, stride u, padding (k-u)/2) is a 3-tap conv whose output channels are (phase, c_out): its
    [L, u*c_out] result IS the upsampled [L*u, c_out] sequence;
  * bias, the leaky_relu between the two convs of a resblock pair, the residual add and the final tanh live in GEMM
    epilogues; the pre-activation of a residual stream and the resblock average are `styler_leaky_sum`.
There is no CPU fallback."""
import torch
import torch.nn as nn

from . import ops

LRELU_SLOPE = 0.1


class AttrDict(dict):
    """hifigan/__init__.py:4-7."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


def config_v1():
    """The generator fields of hifigan/config.json (the universal / LJSpeech V1 generator the reference loads,
    utils.py:251-258)."""
    return AttrDict(resblock="1", upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4],
                    upsample_initial_channel=512, resblock_kernel_sizes=[3, 7, 11],
                    resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], num_mels=80, hop_size=256,
                    sampling_rate=22050)


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


class _WNConv(nn.Module):
    """Parameter holder of a weight-normed Conv1d / ConvTranspose1d: `weight_g` [C0,1,1], `weight_v` [C0,C1,k], `bias`
    (torch.nn.utils.weight_norm, dim 0); `weight` once the norm is removed."""

    def __init__(self, shape, n_bias, kernel_size, dilation=1, stride=1, transposed=False):
        super().__init__()
        v = torch.empty(shape).normal_(0.0, 0.01)
        self.weight_g = nn.Parameter(v.reshape(shape[0], -1).norm(dim=1).reshape(shape[0], 1, 1))
        self.weight_v = nn.Parameter(v)
        self.bias = nn.Parameter(torch.zeros(n_bias))
        self.kernel_size, self.dilation, self.stride, self.transposed = kernel_size, dilation, stride, transposed

    def folded_weight(self):
        if "weight" in self._parameters:
            return self.weight.detach()
        v = self.weight_v.detach()
        norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1)
        return self.weight_g.detach() * v / norm

    def remove_weight_norm(self):
        if "weight" in self._parameters:
            return
        w = self.folded_weight()
        del self.weight_g, self.weight_v
        self.weight = nn.Parameter(w)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # a checkpoint saved after remove_weight_norm() holds `weight`; one saved before holds weight_g / weight_v
        if prefix + "weight" in state_dict and "weight" not in self._parameters:
            self.remove_weight_norm()
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def version(self):
        return tuple(p._version for p in self.parameters()) + tuple(p.data_ptr() for p in self.parameters())


class ResBlock(nn.Module):
    """hifigan/models.py:19-107 (parameters only; the arithmetic is Generator._resblock)."""

    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.h = h
        self.convs1 = nn.ModuleList([_WNConv((channels, channels, kernel_size), channels, kernel_size, d)
                                     for d in dilation])
        self.convs2 = nn.ModuleList([_WNConv((channels, channels, kernel_size), channels, kernel_size, 1)
                                     for _ in dilation])

    def remove_weight_norm(self):
        for l in list(self.convs1) + list(self.convs2):
            l.remove_weight_norm()


def conv_taps(w, prec):
    """Conv1d weight [n, cin, k] -> list of (kernel-layout weight [n, kw*cin], kw, pad) GEMM calls, k <= 9 in one call,
    9 < k <= 11 split at the centre tap (the second call's copy of it is zero)."""
    n, cin, k = w.shape
    wk = w.permute(0, 2, 1).contiguous()                         # [n, k, cin]
    dt = torch.bfloat16 if prec == ops.PREC_BF16 else torch.float32
    if n % 4:                                                    # conv_post: n = 1 -> zero rows up to 4
        wk = torch.cat([wk, wk.new_zeros(4 - n % 4, k, cin)], 0)
    if k <= 9:
        return [(wk.reshape(wk.shape[0], k * cin).to(dt).contiguous(), k, (k - 1) // 2)]
    assert k % 2 == 1 and k <= 11, k
    c = (k - 1) // 2                                             # centre tap
    ka, kb = c + 1, k - c
    wa = wk[:, :ka]
    wb = wk[:, c:].clone()
    wb[:, 0] = 0
    return [(wa.reshape(wk.shape[0], ka * cin).to(dt).contiguous(), ka, c),
            (wb.reshape(wk.shape[0], kb * cin).to(dt).contiguous(), kb, 0)]


def transposed_taps(w, bias, stride, padding, prec):
    """ConvTranspose1d weight [cin, cout, k] -> (W' [u*cout, kw*cin], bias' [u*cout], kw, pad):
    y[t*u + p, co] = sum_i x[i, :] . w[:, co, u*(t - i) + p + padding]  =  a kw-tap conv over t whose tap j reads
    x[t + j - pad] (i.e. t - i = pad - j) and whose output channel index is p*cout + co."""
    cin, cout, k = w.shape
    u = stride
    d_max = (k - 1 - padding) // u
    d_min = -((u - 1 + padding) // u)
    kw = d_max - d_min + 1
    wp = w.new_zeros(u, cout, kw, cin)
    for j in range(kw):
        delta = d_max - j
        for p in range(u):
            kk = u * delta + p + padding
            if 0 <= kk < k:
                wp[p, :, j, :] = w[:, :, kk].t()
    dt = torch.bfloat16 if prec == ops.PREC_BF16 else torch.float32
    return wp.reshape(u * cout, kw * cin).to(dt).contiguous(), bias.repeat(u).contiguous(), kw, d_max


class Generator(nn.Module):
    """hifigan/models.py:112-173."""

    def __init__(self, h):
        super().__init__()
        self.h = h
        self.num_kernels = len(h.resblock_kernel_sizes)
        self.num_upsamples = len(h.upsample_rates)
        c0 = h.upsample_initial_channel
        self.conv_pre = _WNConv((c0, 80, 7), c0, 7)
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
            self.ups.append(_WNConv((c0 // (2 ** i), c0 // (2 ** (i + 1)), k), c0 // (2 ** (i + 1)), k, stride=u,
                                    transposed=True))
        self.resblocks = nn.ModuleList()
        for i in range(len(self.ups)):
            ch = c0 // (2 ** (i + 1))
            for k, d in zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes):
                self.resblocks.append(ResBlock(h, ch, k, d))
        self.conv_post = _WNConv((1, ch, 7), 1, 7)
        self.prec = None                 # None: follow runtime.rt.prec; ops.PREC_F32 / ops.PREC_BF16 to pin
        self.use_graph = False           # replay one captured hipGraph per (B, T): the eager pass is ~250 launches of a
                                         # few microseconds per utterance, i.e. bound by the host's enqueue rate
        self.fork_streams = True         # the three resblocks of a stage on three HIP streams
        self._plan = None
        self._graphs = {}
        self._side = []

    def remove_weight_norm(self):
        for l in self.ups:
            l.remove_weight_norm()
        for l in self.resblocks:
            l.remove_weight_norm()
        self.conv_pre.remove_weight_norm()
        self.conv_post.remove_weight_norm()

    # ---- derived weights (rebuilt when a parameter changes or moves) -------------------------------------------------
    def _convs(self):
        yield "conv_pre", self.conv_pre
        for i, l in enumerate(self.ups):
            yield f"ups.{i}", l
        for j, rb in enumerate(self.resblocks):
            for m, l in enumerate(rb.convs1):
                yield f"resblocks.{j}.convs1.{m}", l
            for m, l in enumerate(rb.convs2):
                yield f"resblocks.{j}.convs2.{m}", l
        yield "conv_post", self.conv_post

    def _prepare(self, prec):
        key = (prec,) + tuple(l.version() for _, l in self._convs())
        if self._plan is not None and self._plan[0] == key:
            return self._plan[1]
        plan = {}
        with torch.no_grad():
            for name, l in self._convs():
                w = l.folded_weight().float()
                b = l.bias.detach().float()
                if l.transposed:
                    wt, bt, kw, pad = transposed_taps(w, b, l.stride, (l.kernel_size - l.stride) // 2, prec)
                    plan[name] = ([(wt, kw, pad)], bt)
                else:
                    if b.numel() % 4:
                        b = torch.cat([b, b.new_zeros(4 - b.numel() % 4)])
                    plan[name] = (conv_taps(w, prec), b.contiguous())
        self._plan = (key, plan)
        self._graphs = {}                # captured graphs hold pointers into the previous plan
        return plan

    # ---- arithmetic ---------------------------------------------------------------------------------------------------
    @staticmethod
    def _conv(x, entry, prec, dilation=1, act=ops.ACT_NONE, res=None, out=None):
        """x [L, cin] contiguous -> [L, n]; `act` is applied before `res` is added (GEMM epilogue order).  A split
        (two-call) conv can fuse neither an activation nor more than its own accumulation into the epilogue, so the
        caller passes act only for single-call convs (see _resblock)."""
        calls, bias = entry
        L = x.shape[0]
        n = calls[0][0].shape[0]
        if out is None:
            out = torch.empty(L, n, device=x.device, dtype=torch.float32)
        assert len(calls) == 1 or act == ops.ACT_NONE
        for r in range(dilation):
            xv = x[r::dilation].unsqueeze(0)
            if xv.shape[1] == 0:
                continue
            ov = out[r::dilation].unsqueeze(0)
            rv = None if res is None else res[r::dilation].unsqueeze(0)
            for c, (w, kw, pad) in enumerate(calls):
                last = c == len(calls) - 1
                ops.conv_gemm_pad(xv, w, bias if last else None, kw=kw, pad=pad, act=act if last else ops.ACT_NONE,
                                  prec=prec, res=rv if c == 0 else ov, out=ov)
        return out

    def _resblock(self, x, plan, j, prec):
        """ResBlock.forward, hifigan/models.py:94-101: x = c2(lrelu(c1(lrelu(x)))) + x for the three dilations."""
        rb = self.resblocks[j]
        for m, l in enumerate(rb.convs1):
            e1, e2 = plan[f"resblocks.{j}.convs1.{m}"], plan[f"resblocks.{j}.convs2.{m}"]
            xt = ops.leaky_sum(x, slope=LRELU_SLOPE)
            if len(e1[0]) == 1:
                t = self._conv(xt, e1, prec, dilation=l.dilation, act=ops.ACT_LEAKY)
            else:
                t = self._conv(xt, e1, prec, dilation=l.dilation)
                ops.leaky_sum(t, slope=LRELU_SLOPE, out=t)
            x = self._conv(t, e2, prec, res=x, out=xt)           # xt is dead: reuse it for the new residual stream
        return x

    def _resblocks(self, x, plan, i, prec):
        """The num_kernels resblocks of a stage are independent (hifigan/models.py:158-163) and, at utterance lengths,
        each of their GEMMs fills a fraction of the 256 CUs: run them on forked HIP streams (inside a hipGraph
        capture they become parallel branches) and join before the average."""
        nk = self.num_kernels
        if not self.fork_streams or nk == 1:
            return [self._resblock(x, plan, i * nk + j, prec) for j in range(nk)]
        cur = torch.cuda.current_stream(x.device)
        if len(self._side) < nk - 1:
            self._side = [torch.cuda.Stream(device=x.device) for _ in range(nk - 1)]
        rs = [None] * nk
        for j in range(nk - 1, 0, -1):                           # the widest kernel size first
            side = self._side[j - 1]
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                rs[j] = self._resblock(x, plan, i * nk + j, prec)
        rs[0] = self._resblock(x, plan, i * nk, prec)
        for side in self._side[:nk - 1]:
            cur.wait_stream(side)                                # (the next fork waits for `cur` again before any
        return rs                                                #  side-stream block can be reused)

    def _item(self, mel, plan, prec):
        """mel [T, 80] -> wav [T * prod(rates)]; hifigan/models.py:155-169."""
        x = self._conv(mel, plan["conv_pre"], prec)
        x = ops.leaky_sum(x, slope=LRELU_SLOPE, out=x)
        for i in range(self.num_upsamples):
            entry = plan[f"ups.{i}"]
            cout = entry[1].numel() // self.h.upsample_rates[i]
            x = self._conv(x, entry, prec).view(-1, cout)        # [L, u*cout] == [L*u, cout]
            rs = self._resblocks(x, plan, i, prec)
            assert 1 <= len(rs) <= 3, "styler_leaky_sum folds at most three resblocks"
            rs += [None] * (3 - len(rs))
            last = i == self.num_upsamples - 1                   # F.leaky_relu(x) before conv_post: default slope 0.01
            x = ops.leaky_sum(rs[0], rs[1], rs[2], scale=1.0 / self.num_kernels, slope=0.01 if last else LRELU_SLOPE)
        y = self._conv(x, plan["conv_post"], prec, act=ops.ACT_TANH)
        return y[:, 0]

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("styler_amd.hifigan.Generator runs on the MI355X HIP path only (no CPU fallback)")
        if self.conv_post.bias.device != x.device:
            raise RuntimeError(f"vocoder weights on {self.conv_post.bias.device}, mel on {x.device}: call .to(device) first")
        if x.dim() == 2:
            x = x.unsqueeze(0)
        if x.dim() != 3 or x.shape[1] != 80:
            raise ValueError(f"expected mel [B, 80, T], got {tuple(x.shape)}")
        prec = self.prec
        if prec is None:
            from .runtime import rt
            prec = rt.kernel_prec()
        plan = self._prepare(prec)
        with torch.no_grad():
            if self.use_graph:
                return self._replay(x, plan, prec)
            return self._run(x, plan, prec)

    def _run(self, x, plan, prec):
        mel = x.float().transpose(1, 2).contiguous()             # [B, T, 80]
        wavs = [self._item(mel[b], plan, prec) for b in range(mel.shape[0])]
        return torch.stack(wavs, 0).unsqueeze(1)

    def _replay(self, x, plan, prec):
        key = (tuple(x.shape), x.dtype, x.device, prec)
        hit = self._graphs.get(key)
        if hit is None:
            static_in = x.clone()
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side):                        # warm the allocator outside the capture
                self._run(static_in, plan, prec)
            torch.cuda.current_stream(x.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self._run(static_in, plan, prec)
            hit = self._graphs[key] = (graph, static_in, static_out)
        graph, static_in, static_out = hit
        static_in.copy_(x)
        graph.replay()
        return static_out.clone()
