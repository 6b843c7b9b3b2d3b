"""CPU: the SecretDecoder's parameter inventory against torchvision's PUBLISHED EfficientNet-B1 (utils/models.py:84-96 builds
``efficientnet_b1`` and swaps ``classifier[1]``).  torchvision is not in the image, so the inventory is re-derived here from its
public construction rule, independently of aqualora_amd/decoder.py's own stage table: the B0 base configuration
(expand, kernel, stride, in, out, layers) scaled by width 1.0 / depth 1.1 with ``ceil`` (torchvision
``_efficientnet_conf("efficientnet_b1")``), ``Conv2dNormActivation`` = [conv (no bias), BatchNorm2d], ``MBConv.block`` =
[expand (absent at ratio 1), depthwise, SqueezeExcitation(fc1, fc2 with squeeze = max(1, block input // 4)), project], head
``Conv2dNormActivation(320, 1280, 1)``, ``classifier = [Dropout, Linear]``.  Pins: the documented parameter count of
``efficientnet_b1`` (7 794 184 with the 1000-way head), and from it the reference decoder's 6 636 160 with the 96-way head."""
import math

from aqualora_amd.decoder import SecretDecoder

B0 = [(1, 3, 1, 32, 16, 1), (6, 3, 2, 16, 24, 2), (6, 5, 2, 24, 40, 2), (6, 3, 2, 40, 80, 3), (6, 5, 1, 80, 112, 3),
      (6, 5, 2, 112, 192, 4), (6, 3, 1, 192, 320, 1)]


def _adjust(c, width):          # torchvision _make_divisible(c * width, 8)
    v = c * width
    new = max(8, int(v + 4) // 8 * 8)
    return new + 8 if new < 0.9 * v else new


def published_b1_inventory(num_out, width=1.0, depth=1.1):
    inv = []

    def cna(prefix, cin, cout, k, groups=1):
        inv.append((f"{prefix}.0.weight", (cout, cin // groups, k, k)))
        inv.append((f"{prefix}.1.weight", (cout,)))
        inv.append((f"{prefix}.1.bias", (cout,)))

    cna("model.features.0", 3, _adjust(32, width), 3)
    for si, (t, k, s, cin, cout, n) in enumerate(B0, start=1):
        cin, cout, n = _adjust(cin, width), _adjust(cout, width), int(math.ceil(n * depth))
        for bi in range(n):
            ci = cin if bi == 0 else cout
            cexp = _adjust(ci * t, 1.0)
            p = f"model.features.{si}.{bi}.block"
            j = 0
            if cexp != ci:
                cna(f"{p}.{j}", ci, cexp, 1)
                j += 1
            cna(f"{p}.{j}", cexp, cexp, k, groups=cexp)
            j += 1
            sq = max(1, ci // 4)
            inv += [(f"{p}.{j}.fc1.weight", (sq, cexp, 1, 1)), (f"{p}.{j}.fc1.bias", (sq,)),
                    (f"{p}.{j}.fc2.weight", (cexp, sq, 1, 1)), (f"{p}.{j}.fc2.bias", (cexp,))]
            j += 1
            cna(f"{p}.{j}", cexp, cout, 1)
    last = _adjust(320, width)
    cna("model.features.8", last, 4 * last, 1)
    inv += [("model.classifier.1.weight", (num_out, 4 * last)), ("model.classifier.1.bias", (num_out,))]
    return inv


def _count(inv):
    return sum(math.prod(s) for _, s in inv)


def test_decoder_parameters_equal_the_published_efficientnet_b1_inventory():
    assert _count(published_b1_inventory(1000)) == 7_794_184          # torchvision's documented efficientnet_b1 size
    want = published_b1_inventory(96)                                 # SecretDecoder(48): classifier[1] = Linear(1280, 96)
    assert _count(want) == 6_636_160 and len(want) == 301
    dec = SecretDecoder(48)
    got = [(n, tuple(p.shape)) for n, p in dec.named_parameters()]
    assert got == want                                                # names, shapes AND registration order
    # BatchNorm buffers ride along in msgdecoder.pt: one (mean, var, counter) triple per normalisation layer
    bn = [n for n, _ in want if n.endswith(".1.weight") and "classifier" not in n]
    bufs = dict(dec.named_buffers())
    assert len(bufs) == 3 * len(bn)
    for n in bn:
        base = n[:-len("weight")]
        assert {base + "running_mean", base + "running_var", base + "num_batches_tracked"} <= set(bufs)


def test_captured_graphs_are_not_part_of_a_module_copy():
    """copy.deepcopy / pickle of the decoder or the U-Net must not trip over the derived GPU state they keep (folded weights with
    their HIP graphs, captured sampling loops): `__getstate__` drops it and the copy rebuilds it on first use."""
    import copy

    import torch
    from aqualora_amd.decoder import SecretDecoder
    from tests.common import tiny_unet

    class NoCopy:
        def __deepcopy__(self, memo):
            raise TypeError("a HIP graph cannot be copied")

    dec = SecretDecoder(8)
    dec._packed = {"graphs": NoCopy()}
    assert copy.deepcopy(dec)._packed is None and dec._packed is not None
    unet = tiny_unet("cpu", torch.float32)
    unet.__dict__["_aql_loops"] = {"loop": NoCopy()}
    assert "_aql_loops" not in copy.deepcopy(unet).__dict__ and "_aql_loops" in unet.__dict__
