"""Algorithmic work (FLOPs, HBM bytes) of every kernel launch in the native D-step / G-step.

Pure arithmetic, no device code: bench.py prices each launch with these figures (roofline),
tools/layer_spec.py prints the table quoted in DESIGN.md / SURVEY.md 8(d).

Byte model (layer-granular, fp32): every conv reads its input once and writes its output once;
a fused residual / gradient add reads one more activation-sized tensor; a fused activation
derivative reads the saved activation; weights (and weight grads) are counted once per launch.
FLOPs = 2 * MACs of the dense arithmetic actually required.
"""

G_UPS = ((512, 256, 16, 8, 4), (256, 128, 16, 8, 4), (128, 64, 4, 2, 1), (64, 32, 4, 2, 1))
D_MAIN = ((1, 16, 15, 1, 7, 1), (16, 64, 41, 4, 20, 4), (64, 256, 41, 4, 20, 16),
          (256, 1024, 41, 4, 20, 64), (1024, 1024, 41, 4, 20, 256), (1024, 1024, 5, 1, 2, 1))
F32 = 4


def conv_out_len(lin, k, stride, pad, dil=1):
    return (lin + 2 * pad - dil * (k - 1) - 1) // stride + 1


def conv_cost(B, Cin, Lin, Cout, K, stride, pad, dil, groups, which, extra_reads=0, extra_writes=0,
              act_read=False):
    """which: 'fwd' | 'bwd_data' | 'bwd_weight'.  extra_reads/extra_writes: additional
    activation-sized tensors of the OUTPUT side of that launch (residual, y_act, gx_add)."""
    Lout = conv_out_len(Lin, K, stride, pad, dil)
    macs = B * Cout * Lout * (Cin // groups) * K
    nin, nout, nw = B * Cin * Lin, B * Cout * Lout, Cout * (Cin // groups) * K + Cout
    if which == "fwd":
        elems = nin + nout + nw + (extra_reads + extra_writes) * nout
    elif which == "bwd_data":   # reads gy (+y_act), writes gx (+gx_add read)
        elems = nout + (nout if act_read else 0) + nin + nw + extra_reads * nin
    else:                        # bwd_weight: reads x, gy (+y_act), writes gw
        elems = nin + nout + (nout if act_read else 0) + nw
    return {"flops": 2 * macs, "bytes": F32 * elems}


def convt_as_conv(B, Cin, Lin, Cout, K, stride, pad):
    """Geometry of the mirrored conv whose backward-data is the transposed conv."""
    Lout = (Lin - 1) * stride - 2 * pad + K
    return dict(B=B, Cin=Cout, Lin=Lout, Cout=Cin, K=K, stride=stride, pad=pad, dil=1, groups=1)


def _atom(B, C, L, d, mode, out):
    c0 = dict(B=B, Cin=C, Lin=L, Cout=C, K=3, stride=1, pad=d, dil=d, groups=1)
    c1 = dict(B=B, Cin=C, Lin=L, Cout=C, K=3, stride=1, pad=1, dil=1, groups=1)
    if mode == "fwd":
        out.append(("atom%d.conv0.fwd" % C, conv_cost(which="fwd", **c0)))
        out.append(("atom%d.conv1.fwd" % C, conv_cost(which="fwd", extra_reads=1, **c1)))
    elif mode == "fwd_train":   # also writes the pre-residual activation for the LeakyReLU mask
        out.append(("atom%d.conv0.fwd" % C, conv_cost(which="fwd", **c0)))
        out.append(("atom%d.conv1.fwd" % C, conv_cost(which="fwd", extra_reads=1, extra_writes=1, **c1)))
    else:
        out.append(("atom%d.conv1.bwd_weight" % C, conv_cost(which="bwd_weight", act_read=True, **c1)))
        out.append(("atom%d.conv1.bwd_data" % C, conv_cost(which="bwd_data", act_read=True, **c1)))
        out.append(("atom%d.conv0.bwd_weight" % C, conv_cost(which="bwd_weight", act_read=True, **c0)))
        out.append(("atom%d.conv0.bwd_data" % C, conv_cost(which="bwd_data", act_read=True, extra_reads=1, **c0)))


def generator_launches(B, mels, T, mode):
    """mode: 'fwd' (inference / D-step), 'fwd_train' (G-step forward), 'bwd'."""
    out = []
    L = T
    first = dict(B=B, Cin=mels, Lin=L, Cout=512, K=7, stride=1, pad=3, dil=1, groups=1)
    stages = []
    for cin, cout, k, s, p in G_UPS:
        stages.append((cin, cout, k, s, p, L))
        L = (L - 1) * s - 2 * p + k
    last = dict(B=B, Cin=32, Lin=L, Cout=1, K=7, stride=1, pad=3, dil=1, groups=1)
    if mode != "bwd":
        out.append(("g.first.fwd", conv_cost(which="fwd", **first)))
        for cin, cout, k, s, p, lin in stages:
            m = convt_as_conv(B, cin, lin, cout, k, s, p)
            out.append(("g.convT%d.fwd" % cout, conv_cost(which="bwd_data", **m)))
            for d in (1, 3, 9):
                _atom(B, cout, m["Lin"], d, mode, out)
        out.append(("g.last.fwd", conv_cost(which="fwd", **last)))
    else:
        out.append(("g.last.bwd_weight", conv_cost(which="bwd_weight", act_read=True, **last)))
        out.append(("g.last.bwd_data", conv_cost(which="bwd_data", act_read=True, **last)))
        for cin, cout, k, s, p, lin in reversed(stages):
            m = convt_as_conv(B, cin, lin, cout, k, s, p)
            for d in (9, 3, 1):
                _atom(B, cout, m["Lin"], d, "bwd", out)
            out.append(("g.convT%d.bwd_weight" % cout, conv_cost(which="bwd_weight", act_read=True, **m)))
            out.append(("g.convT%d.bwd_data" % cout, conv_cost(which="fwd", extra_reads=1, **m)))
        out.append(("g.first.bwd_weight", conv_cost(which="bwd_weight", act_read=True, **first)))
    return out


def discriminator_launches(B, L0, mode, need_gx=False, feat_grads=False):
    """mode 'fwd' | 'bwd' over the 3 scales (the pooling kernels are priced as pure streams)."""
    out = []
    L = L0
    for s in range(3):
        if s:
            lp = (L + 4 - 4) // 2 + 1
            out.append(("d.pool.%s" % mode, {"flops": 4 * B * lp, "bytes": F32 * B * (L + lp)}))
            L = lp
        l = L
        geo = []
        for cin, cout, k, st, p, g in D_MAIN:
            geo.append(dict(B=B, Cin=cin, Lin=l, Cout=cout, K=k, stride=st, pad=p, dil=1, groups=g))
            l = conv_out_len(l, k, st, p)
        judge = dict(B=B, Cin=1024, Lin=l, Cout=1, K=3, stride=1, pad=1, dil=1, groups=1)
        if mode == "fwd":
            for i, c in enumerate(geo):
                out.append(("d.main%d.fwd" % i, conv_cost(which="fwd", **c)))
            out.append(("d.judge.fwd", conv_cost(which="fwd", **judge)))
        else:
            out.append(("d.judge.bwd_data", conv_cost(which="bwd_data", extra_reads=int(feat_grads), **judge)))
            for i in range(5, -1, -1):
                if not need_gx or True:
                    pass
                if i > 0 or need_gx:
                    out.append(("d.main%d.bwd_data" % i,
                                conv_cost(which="bwd_data", act_read=True,
                                          extra_reads=int(feat_grads and i > 0), **geo[i])))
    return out


def discriminator_wgrad_launches(B, L0):
    out = []
    L = L0
    for s in range(3):
        if s:
            L = (L + 4 - 4) // 2 + 1
        l = L
        for i, (cin, cout, k, st, p, g) in enumerate(D_MAIN):
            out.append(("d.main%d.bwd_weight" % i,
                        conv_cost(B, cin, l, cout, k, st, p, 1, g, "bwd_weight", act_read=True)))
            l = conv_out_len(l, k, st, p)
        out.append(("d.judge.bwd_weight", conv_cost(B, 1024, l, 1, 3, 1, 1, 1, 1, "bwd_weight")))
    return out


def feature_elems(B, L0):
    tot, L = 0, L0
    for s in range(3):
        if s:
            L = (L + 4 - 4) // 2 + 1
        l = L
        for cin, cout, k, st, p, g in D_MAIN:
            l = conv_out_len(l, k, st, p)
            tot += B * cout * l
    return tot


def d_step_launches(B, mels=80, T=32):
    L0 = T * 256
    out = generator_launches(B, mels, T, "fwd")
    for _ in range(2):   # fake pass, real pass
        out += discriminator_launches(B, L0, "fwd")
    for _ in range(2):
        out += discriminator_launches(B, L0, "bwd", need_gx=False)
        out += discriminator_wgrad_launches(B, L0)
    nparam = 5637953
    out.append(("adam.D", {"flops": 12 * nparam, "bytes": 7 * F32 * nparam}))
    return out


def g_step_launches(B, mels=80, T=32):
    L0 = T * 256
    out = generator_launches(B, mels, T, "fwd_train")
    out += discriminator_launches(B, L0, "fwd")
    out += discriminator_launches(B, L0, "fwd")
    fe = feature_elems(B, L0)
    out.append(("loss.l1.fwd", {"flops": 3 * fe, "bytes": 2 * F32 * fe}))
    out.append(("loss.l1.bwd", {"flops": 2 * fe, "bytes": 3 * F32 * fe}))
    out += discriminator_launches(B, L0, "bwd", need_gx=True, feat_grads=True)
    out += generator_launches(B, mels, T, "bwd")
    nparam = 4519937 + (mels - 80) * 512 * 7
    out.append(("adam.G", {"flops": 12 * nparam, "bytes": 7 * F32 * nparam}))
    return out


# ---- the byte model of the program AS SCHEDULED (r05): the layer-granular model above prices every conv as a launch of its
# own (SURVEY.md 8(d)'s contract: 96 atom convs per pair reading and writing whole tensors).  What runs fuses each ResidualAtom
# into one launch per pass (csrc/atom_fused.hip: the value between its two convs never leaves the chip), whole stacks into one
# launch where nothing is saved (csrc/stack_fused.hip, <= 64 channels), keeps the LeakyReLU masks of a training atom as one bit
# per element (sign words: 1 / 32 of an fp32 tensor each for u and t), runs the discriminator once over [fake; real] in the
# D-step and fuses the 18-map L1 loss with its gradient.  Same FLOPs, fewer necessary bytes.
SIGN_WORDS = 1.0 / 32        # one sign-word tensor relative to the fp32 tensor it stands for
STACK_FUSED_MAX_C = 64       # P.stack_supported: the one-launch inference stack takes 32 / 64 channels


def _atom_fused(B, C, L, d, mode, out):
    n = B * C * L
    c0 = conv_cost(B, C, L, C, 3, 1, d, d, 1, "fwd")
    fl, wb = 2 * c0["flops"], F32 * 2 * (3 * C * C + C)
    if mode == "fwd":            # x in, y out
        out.append(("atom%d.fwd" % C, {"flops": fl, "bytes": int(F32 * n * 2) + wb}))
    elif mode == "fwd_train":    # x in; y, t out; sign words of u and t
        out.append(("atom%d.fwd_train" % C, {"flops": fl, "bytes": int(F32 * n * (3 + 2 * SIGN_WORDS)) + wb}))
    else:                        # g in, sign words of u and t in; gt, gx out -- then the two weight gradients
        out.append(("atom%d.bwd_data" % C, {"flops": fl, "bytes": int(F32 * n * (3 + 2 * SIGN_WORDS)) + wb}))
        for nm in ("conv1", "conv0"):       # operands (t, g) / (x, gt) + one sign-word tensor; gw out
            out.append(("atom%d.%s.bwd_weight" % (C, nm),
                        {"flops": c0["flops"], "bytes": int(F32 * n * (2 + SIGN_WORDS)) + wb // 2}))


def fused_generator_launches(B, mels, T, mode):
    """generator_launches with the atoms / stacks as the fused launches that run."""
    out = []
    for name, c in generator_launches(B, mels, T, mode):
        if not name.startswith("atom"):
            out.append((name, c))
    # atoms: same walk as generator_launches
    L = T
    atoms = []
    for cin, cout, k, s, p in G_UPS:
        L = (L - 1) * s - 2 * p + k
        if mode == "fwd" and cout <= STACK_FUSED_MAX_C:
            n = B * cout * L
            fl = sum(2 * conv_cost(B, cout, L, cout, 3, 1, d, d, 1, "fwd")["flops"] for d in (1, 3, 9))
            atoms.append(("stack%d.fwd" % cout, {"flops": fl, "bytes": F32 * n * 2 + 3 * F32 * 2 * (3 * cout * cout + cout)}))
            continue
        for d in ((1, 3, 9) if mode != "bwd" else (9, 3, 1)):
            _atom_fused(B, cout, L, d, mode, atoms)
    return out + atoms


def fused_d_step_launches(B, mels=80, T=32):
    L0 = T * 256
    out = fused_generator_launches(B, mels, T, "fwd")
    out += discriminator_launches(2 * B, L0, "fwd")                      # one pass over [fake; real]
    out += discriminator_launches(2 * B, L0, "bwd", need_gx=False)
    out += discriminator_wgrad_launches(2 * B, L0)
    nparam = 5637953
    out.append(("adam.D", {"flops": 12 * nparam, "bytes": 7 * F32 * nparam}))
    return out


def fused_g_step_launches(B, mels=80, T=32):
    L0 = T * 256
    out = fused_generator_launches(B, mels, T, "fwd_train")
    out += discriminator_launches(B, L0, "fwd")
    out += discriminator_launches(B, L0, "fwd")
    fe = feature_elems(B, L0)
    out.append(("loss.l1.fwd_bwd", {"flops": 5 * fe, "bytes": 3 * F32 * fe}))      # r, f in; d loss / d f out: one pass
    out += discriminator_launches(B, L0, "bwd", need_gx=True, feat_grads=True)
    out += fused_generator_launches(B, mels, T, "bwd")
    nparam = 4519937 + (mels - 80) * 512 * 7
    out.append(("adam.G", {"flops": 12 * nparam, "bytes": 7 * F32 * nparam}))
    return out


SURVEY_MB_PER_ELEMENT_PER_CALL = 135.0      # mean necessary bytes per batch element per trainer call


# ---- the weight-normed MelGAN (experiment/realmelgan.py:15-181): same discriminator geometry, a generator of
# ResnetBlocks -- shortcut1x1(x) + conv1x1(lrelu(conv_k3_dil(reflpad(lrelu(x))))) -- instead of ResidualAtoms.  Weight
# normalisation (w = g v / |v| per output channel) is O(parameters) per pass and not priced.
REAL_UPS = ((512, 256, 16, 8, 4), (256, 128, 16, 8, 4), (128, 64, 4, 2, 1), (64, 32, 4, 2, 1))


def _resblock(B, C, L, d, mode, out):
    sc = dict(B=B, Cin=C, Lin=L, Cout=C, K=1, stride=1, pad=0, dil=1, groups=1)
    c3 = dict(B=B, Cin=C, Lin=L, Cout=C, K=3, stride=1, pad=d, dil=d, groups=1)
    if mode in ("fwd", "fwd_train"):
        out.append(("res%d.shortcut.fwd" % C, conv_cost(which="fwd", **sc)))
        out.append(("res%d.conv3.fwd" % C, conv_cost(which="fwd", **c3)))
        out.append(("res%d.conv1.fwd" % C, conv_cost(which="fwd", extra_reads=1, **sc)))
    else:
        out.append(("res%d.conv1.bwd_weight" % C, conv_cost(which="bwd_weight", **sc)))
        out.append(("res%d.conv1.bwd_data" % C, conv_cost(which="bwd_data", **sc)))
        out.append(("res%d.conv3.bwd_weight" % C, conv_cost(which="bwd_weight", act_read=True, **c3)))
        out.append(("res%d.conv3.bwd_data" % C, conv_cost(which="bwd_data", act_read=True, **c3)))
        out.append(("res%d.shortcut.bwd_weight" % C, conv_cost(which="bwd_weight", **sc)))
        out.append(("res%d.shortcut.bwd_data" % C, conv_cost(which="bwd_data", extra_reads=1, **sc)))


def real_generator_launches(B, mels, T, mode):
    """Generator(mels, 32, 3) of experiment/realmelgan.py; mode as generator_launches."""
    out = []
    L = T
    first = dict(B=B, Cin=mels, Lin=L, Cout=512, K=7, stride=1, pad=3, dil=1, groups=1)
    stages = []
    for cin, cout, k, s, p in REAL_UPS:
        stages.append((cin, cout, k, s, p, L))
        L = (L - 1) * s - 2 * p + k
    last = dict(B=B, Cin=32, Lin=L, Cout=1, K=7, stride=1, pad=3, dil=1, groups=1)
    if mode != "bwd":
        out.append(("g.first.fwd", conv_cost(which="fwd", **first)))
        for cin, cout, k, s, p, lin in stages:
            m = convt_as_conv(B, cin, lin, cout, k, s, p)
            out.append(("g.convT%d.fwd" % cout, conv_cost(which="bwd_data", **m)))
            for d in (1, 3, 9):
                _resblock(B, cout, m["Lin"], d, mode, out)
        out.append(("g.last.fwd", conv_cost(which="fwd", **last)))
    else:
        out.append(("g.last.bwd_weight", conv_cost(which="bwd_weight", act_read=True, **last)))
        out.append(("g.last.bwd_data", conv_cost(which="bwd_data", act_read=True, **last)))
        for cin, cout, k, s, p, lin in reversed(stages):
            m = convt_as_conv(B, cin, lin, cout, k, s, p)
            for d in (9, 3, 1):
                _resblock(B, cout, m["Lin"], d, "bwd", out)
            out.append(("g.convT%d.bwd_weight" % cout, conv_cost(which="bwd_weight", act_read=True, **m)))
            out.append(("g.convT%d.bwd_data" % cout, conv_cost(which="fwd", extra_reads=1, **m)))
        out.append(("g.first.bwd_weight", conv_cost(which="bwd_weight", **first)))
    return out


def real_generator_nparam(mels):
    """weight_v + weight_g + bias of every weight-normed layer."""
    n = mels * 512 * 7 + 2 * 512
    for cin, cout, k, s, p in REAL_UPS:
        n += cin * cout * k + cin + cout                    # ConvTranspose1d: g per INPUT channel (dim 0 of its weight)
        n += 3 * (cout * cout * 3 + 2 * (cout * cout) + 3 * 2 * cout)
    return n + 32 * 7 + 2


REAL_NPARAM_D = 3 * (5637953 + 16 + 64 + 256 + 1024 + 1024 + 1024 + 1)      # three independent discriminators: the headline D's tensors + one g per output channel


def real_d_step_launches(B, mels=128, T=32):
    L0 = T * 256
    out = real_generator_launches(B, mels, T, "fwd")
    for _ in range(2):
        out += discriminator_launches(B, L0, "fwd")
    for _ in range(2):
        out += discriminator_launches(B, L0, "bwd", need_gx=False)
        out += discriminator_wgrad_launches(B, L0)
    out.append(("adam.D", {"flops": 12 * REAL_NPARAM_D, "bytes": 7 * F32 * REAL_NPARAM_D}))
    return out


def real_g_step_launches(B, mels=128, T=32):
    L0 = T * 256
    out = real_generator_launches(B, mels, T, "fwd_train")
    out += discriminator_launches(B, L0, "fwd")
    out += discriminator_launches(B, L0, "fwd")
    fe = feature_elems(B, L0)
    out.append(("loss.l1.fwd", {"flops": 3 * fe, "bytes": 2 * F32 * fe}))
    out.append(("loss.l1.bwd", {"flops": 2 * fe, "bytes": 3 * F32 * fe}))
    out += discriminator_launches(B, L0, "bwd", need_gx=True, feat_grads=True)
    out += real_generator_launches(B, mels, T, "bwd")
    n = real_generator_nparam(mels)
    out.append(("adam.G", {"flops": 12 * n, "bytes": 7 * F32 * n}))
    return out


# ---- stage 1 (SURVEY.md 8(f) row 2 / BASELINE configs[4]): featuregenerator/upscale.py:85-99, featurediscriminator/upscale.py:7-27
S1_G = ((1024, 512, 4, 2), (512, 256, 4, 2), (256, 128, 4, 2), (128, 128, 4, 2), (128, 64, 4, 2), (64, 32, 3, 1),
        (32, 1, 3, 1))          # (Cin, Cout, kH, sH) of the ConvTranspose2d stack; kW = 4, sW = 2, padding (1, 1); from 4 x 4
S1_D_DIL = (1, 3, 9, 27, 81, 1, 1)
S1_NPARAM_G, S1_NPARAM_D = 13542881, 1278209


def convt2d_cost(B, Cin, H, W, Cout, kH, sH, which, act_read=True):
    """ConvTranspose2d k(kH, 4) s(sH, 2) p(1, 1): every output pixel meets (kH / sH) x 2 taps of every input channel."""
    nin, nout, nw = B * Cin * H * W, B * Cout * (H * sH) * (2 * W), Cin * Cout * kH * 4 + Cout
    macs = B * H * W * Cin * Cout * kH * 4
    if which == "fwd":
        elems = nin + nout + nw
    elif which == "bwd_data":
        elems = nout + (nout if act_read else 0) + nin + nw
    else:
        elems = nin + nout + (nout if act_read else 0) + nw
    return {"flops": 2 * macs, "bytes": F32 * elems}


def stage1_generator_launches(B, mode, noise_dim=128):
    """mode: 'fwd' | 'bwd' (weight gradients of every layer, data gradients down to the Linear's output)."""
    out = []
    lin = {"flops": 2 * B * noise_dim * 16384, "bytes": F32 * (B * noise_dim + B * 16384 + noise_dim * 16384 + 16384)}
    H = W = 4
    geo = []
    for cin, cout, kh, sh in S1_G:
        geo.append((cin, H, W, cout, kh, sh))
        H, W = H * sh, 2 * W
    if mode == "fwd":
        out.append(("s1g.linear.fwd", lin))
        for i, (cin, h, w, cout, kh, sh) in enumerate(geo):
            out.append(("s1g.convT2d%d.fwd" % i, convt2d_cost(B, cin, h, w, cout, kh, sh, "fwd")))
    else:
        for i, (cin, h, w, cout, kh, sh) in reversed(list(enumerate(geo))):
            last = i == len(geo) - 1                     # no activation behind the last layer
            out.append(("s1g.convT2d%d.bwd_weight" % i, convt2d_cost(B, cin, h, w, cout, kh, sh, "bwd_weight", not last)))
            out.append(("s1g.convT2d%d.bwd_data" % i, convt2d_cost(B, cin, h, w, cout, kh, sh, "bwd_data", not last)))
        out.append(("s1g.linear.bwd_weight", lin))
    return out


def stage1_discriminator_launches(B, mode, T=512, feat=128, ch=256):
    """mode: 'fwd' | 'bwd_data' | 'bwd_weight'.  Residual layers read the skip once more (act(z + x))."""
    out = []
    for i, d in enumerate(S1_D_DIL):
        cin = feat if i == 0 else ch
        c = dict(B=B, Cin=cin, Lin=T, Cout=ch, K=3, stride=1, pad=d, dil=d, groups=1)
        if mode == "fwd":
            out.append(("s1d.conv%d.fwd" % i, conv_cost(which="fwd", extra_reads=int(i > 0), **c)))
        elif mode == "bwd_weight":
            out.append(("s1d.conv%d.bwd_weight" % i, conv_cost(which="bwd_weight", act_read=True, **c)))
        else:
            out.append(("s1d.conv%d.bwd_data" % i, conv_cost(which="bwd_data", act_read=True, extra_reads=int(i > 0), **c)))
    j = dict(B=B, Cin=ch, Lin=T, Cout=1, K=1, stride=1, pad=0, dil=1, groups=1)
    out.append(("s1d.judge.%s" % mode, conv_cost(which=mode, **j)))
    return out


def stage1_d_step_launches(B):
    out = stage1_generator_launches(B, "fwd")
    out += stage1_discriminator_launches(2 * B, "fwd")                  # one pass over [fake; real]
    out += stage1_discriminator_launches(2 * B, "bwd_weight")
    out += [l for l in stage1_discriminator_launches(2 * B, "bwd_data") if not l[0].startswith("s1d.conv0.")]
    out.append(("adam.s1D", {"flops": 12 * S1_NPARAM_D, "bytes": 7 * F32 * S1_NPARAM_D}))
    return out


def stage1_g_step_launches(B):
    out = stage1_generator_launches(B, "fwd")
    out += stage1_discriminator_launches(B, "fwd")                      # least-squares generator loss: the fake pass only
    out += stage1_discriminator_launches(B, "bwd_data")
    out += stage1_generator_launches(B, "bwd")
    out.append(("adam.s1G", {"flops": 12 * S1_NPARAM_G, "bytes": 7 * F32 * S1_NPARAM_G}))
    return out


def totals(launches):
    return {"flops": sum(c["flops"] for _, c in launches), "bytes": sum(c["bytes"] for _, c in launches)}


def roofline_seconds(launches, hbm_bytes_per_s, flops_per_s):
    """Sum over launches of max(bytes / BW, flops / peak): the per-layer roofline of the step."""
    return sum(max(c["bytes"] / hbm_bytes_per_s, c["flops"] / flops_per_s) for _, c in launches)
