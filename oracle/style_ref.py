"""AdaIN style-transfer network, CPU oracle (test-only).

  * calc_mean_std_ref <- lib/models/Style_net.py:4-12  (unbiased variance + 1e-5)
  * adain_ref         <- lib/models/Style_net.py:21-29
  * make_vgg_ref / make_decoder_ref <- lib/models/Style_net.py:32-118 (VGG-19 with reflection padding; mirrored decoder)
  * style_forward_ref <- lib/models/Style_net.py:163-177, returning only g_t: the sole output the training loop
    consumes (train_human.py:275,350,355 take [2])
  * gram_matrix_ref   <- lib/models/Style_net.py:14-19
  * style_forward_full_ref <- lib/models/Style_net.py:136-177: (loss_c, loss_s, g_t) with the re-encode of g_t, the content
    MSE on relu4_1 and the four Gram-matrix MSEs (relu1_1 .. relu4_1)
"""
import torch
import torch.nn as nn

VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512]
DEC_CFG = [(512, 256), "U", (256, 256), (256, 256), (256, 256), (256, 128), "U", (128, 128), (128, 64), "U",
           (64, 64), (64, 3)]


def calc_mean_std_ref(feat, eps=1e-5):
    assert feat.dim() == 4
    N, C = feat.shape[:2]
    v = feat.reshape(N, C, -1).var(dim=2) + eps
    return feat.reshape(N, C, -1).mean(dim=2).reshape(N, C, 1, 1), v.sqrt().reshape(N, C, 1, 1)


def adain_ref(content, style):
    assert content.shape[:2] == style.shape[:2]
    sm, ss = calc_mean_std_ref(style)
    cm, cs = calc_mean_std_ref(content)
    return (content - cm) / cs * ss + sm


def make_vgg_ref():
    mods = [nn.Conv2d(3, 3, 1)]
    cin = 3
    for v in VGG_CFG:
        if v == "M":
            mods.append(nn.MaxPool2d(2, 2, 0, ceil_mode=True))
        else:
            mods += [nn.ReflectionPad2d(1), nn.Conv2d(cin, v, 3), nn.ReLU()]
            cin = v
    return nn.Sequential(*mods)


def make_decoder_ref():
    mods = []
    for i, v in enumerate(DEC_CFG):
        if v == "U":
            mods.append(nn.Upsample(scale_factor=2, mode="nearest"))
        else:
            mods += [nn.ReflectionPad2d(1), nn.Conv2d(v[0], v[1], 3)]
            if i != len(DEC_CFG) - 1:
                mods.append(nn.ReLU())
    return nn.Sequential(*mods)


def style_forward_ref(vgg31, decoder, content, style, alpha=1.0):
    assert 0 <= alpha <= 1
    sf = vgg31(style)
    cf = vgg31(content)
    t = adain_ref(cf, sf)
    t = alpha * t + (1 - alpha) * cf
    return decoder(t)


def gram_matrix_ref(y):
    b, ch, h, w = y.shape
    f = y.reshape(b, ch, w * h)
    return f.bmm(f.transpose(1, 2)) / (ch * h * w)


def encode_with_intermediate_ref(vgg31, x):
    """relu1_1, relu2_1, relu3_1, relu4_1: the slices [:4], [4:11], [11:18], [18:31] of the encoder (Style_net.py:126-129)."""
    ch = list(vgg31.children())
    outs = []
    for lo, hi in ((0, 4), (4, 11), (11, 18), (18, 31)):
        for m in ch[lo:hi]:
            x = m(x)
        outs.append(x)
    return outs


def style_forward_full_ref(vgg31, decoder, content, style, alpha=1.0):
    assert 0 <= alpha <= 1
    style_feats = encode_with_intermediate_ref(vgg31, style)
    cf = vgg31(content)
    t = adain_ref(cf, style_feats[-1])
    t = alpha * t + (1 - alpha) * cf
    g_t = decoder(t)
    g_feats = encode_with_intermediate_ref(vgg31, g_t)
    mse = torch.nn.functional.mse_loss
    loss_c = mse(g_feats[-1], t)
    loss_s = sum(mse(gram_matrix_ref(a), gram_matrix_ref(b)) for a, b in zip(g_feats, style_feats))
    return loss_c, loss_s, g_t
