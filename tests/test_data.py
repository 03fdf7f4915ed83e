"""Input pipeline on the device (data_loader.py:74-82, 95-100, 113-133): uegan_input_transform against Pillow
itself (the reference's resampler: bit-exact), and DeviceLoader against the per-image oracle with the documented draw order."""
import numpy as np
import pytest
import torch

from helpers import BACKENDS, use_backend
from oracle import uegan_oracle as O
from uegan_amd import data


def _rgb(seed, h, w):
    g = np.random.default_rng(seed)
    # smooth gradients + noise + saturated patches: exercises rounding, the clip and every tap
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(yy * 255 / max(h - 1, 1)), (xx * 255 / max(w - 1, 1)), ((yy + xx) % 256)], -1)
    img = base + g.integers(-40, 40, size=(h, w, 3))
    img[: h // 4, : w // 5] = 255
    img[h // 2:, w // 2:] = g.integers(0, 256, size=(h - h // 2, w - w // 2, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def test_resample_table_properties():
    tab, k = data.resample_table(512, 256)               # the default configuration's 2x reduction
    assert k == 5 and tab.shape == (256, 7)
    assert (tab[:, 2:].sum(1) - (1 << data.PRECISION_BITS)).__abs__().max() <= 2      # normalised coefficients
    assert tab[0, 0] == 0 and tab[-1, 0] + tab[-1, 1] == 512
    tab, k = data.resample_table(20, 20)                 # same size: the identity
    assert all(tab[i, 0] == i and tab[i, 2] == 1 << data.PRECISION_BITS and tab[i, 3:].sum() == 0 for i in range(20))


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("crop,resize", [(40, 20), (37, 20), (20, 20), (20, 33), (48, 13)])
def test_train_transform_matches_pillow(backend, crop, resize):
    dev = use_backend(backend)
    H, W = crop + 11, crop + 7
    imgs = [_rgb(10 + i, H, W) for i in range(5)]
    wins = [(3, 2, 0), (0, 0, 1), (11, 7, 2), (5, 1, 3), (1, 6, 3)]
    pix = torch.from_numpy(np.stack([im[t:t + crop, l:l + crop] for im, (t, l, _) in zip(imgs, wins)])).to(dev)
    out = data.input_transform(pix, resize, [b for _, _, b in wins])
    assert out.shape == (5, 3, resize, resize) and out.dtype == torch.float32
    for i, (im, (t, l, b)) in enumerate(zip(imgs, wins)):
        ref = O.train_transform(im, t, l, crop, resize, b)
        assert torch.equal(out[i].cpu(), ref), "image %d: max diff %g" % (i, float((out[i].cpu() - ref).abs().max()))


@pytest.mark.parametrize("backend", BACKENDS)
def test_test_transform_matches_pillow(backend):
    dev = use_backend(backend)
    for seed, (h, w, s) in enumerate([(31, 45, 24), (50, 29, 32), (24, 24, 24), (17, 23, 40)]):
        im = _rgb(seed, h, w)
        out = data.input_transform(torch.from_numpy(im)[None].to(dev), s)
        assert torch.equal(out[0].cpu(), O.test_transform(im, s))
    with pytest.raises(ValueError):
        data.input_transform(torch.zeros((1, 8, 8, 4), dtype=torch.uint8, device=dev), 8)


def test_draw_order():
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    top, left, bits = data.draw_train_params(30, 40, 16, g1)
    e_top = int(torch.randint(0, 15, size=(1,), generator=g2))
    e_left = int(torch.randint(0, 25, size=(1,), generator=g2))
    e_bits = (1 if float(torch.rand(1, generator=g2)) < 0.5 else 0) | (2 if float(torch.rand(1, generator=g2)) < 0.5 else 0)
    assert (top, left, bits) == (e_top, e_left, e_bits)
    # an image that already has the crop size draws no window (RandomCrop.get_params returns (0, 0) early)
    g1, g2 = torch.Generator().manual_seed(6), torch.Generator().manual_seed(6)
    assert data.draw_train_params(16, 16, 16, g1)[:2] == (0, 0)
    torch.rand(2, generator=g2)
    assert torch.equal(torch.rand(1, generator=g1), torch.rand(1, generator=g2))
    with pytest.raises(ValueError):
        data.draw_train_params(10, 40, 16)


def _make_tree(root, n, sizes):
    from PIL import Image
    (root / "exp").mkdir()
    (root / "raw").mkdir()
    arrs = {}
    for i in range(n):
        h, w = sizes[i % len(sizes)]
        for d, ext in (("exp", "png"), ("raw", "png")):
            a = _rgb(100 * (d == "raw") + i, h, w)
            Image.fromarray(a, "RGB").save(root / d / ("im%02d.%s" % (i, ext)))
            arrs[(d, i)] = a
    return arrs


@pytest.mark.parametrize("backend", BACKENDS)
def test_device_loader_train_and_test(backend, tmp_path):
    use_backend(backend)
    arrs = _make_tree(tmp_path, 7, [(40, 52), (36, 36), (61, 33)])
    ds = data.ReferenceDataset(str(tmp_path))
    assert len(ds) == 7
    files = sorted(p.name for p in (tmp_path / "exp").iterdir())
    # listing order is rglob order (unsorted, like the reference's ReferenceDataset); map sample -> array by file name
    def arr_of(path):
        return arrs[(path.parent.name, int(path.stem[2:]))]
    assert sorted(ds[i][0].name for i in range(7)) == files

    loader = data.get_train_loader(str(tmp_path), img_size=32, resize_size=16, batch_size=3, shuffle=True, num_workers=3,
                                   generator=torch.Generator().manual_seed(11))
    assert len(loader) == 2                                            # drop_last
    g = torch.Generator().manual_seed(11)
    order = torch.randperm(7, generator=g).tolist()
    got = list(loader)
    assert len(got) == 2
    for bi, batch in enumerate(got):
        assert batch.img_exp.shape == (3, 3, 16, 16) and batch.img_raw.shape == (3, 3, 16, 16)
        for k, idx in enumerate(order[3 * bi:3 * bi + 3]):
            a, b, name = ds[idx]
            assert batch.img_name[k] == name == b.stem
            for path, t in ((a, batch.img_exp[k]), (b, batch.img_raw[k])):
                im = arr_of(path)
                top, left, bits = data.draw_train_params(im.shape[0], im.shape[1], 32, g)
                assert torch.equal(t.cpu(), O.train_transform(im, top, left, 32, 16, bits))
    # InputFetcher: endless, restarts the loader (a new permutation) when it runs out
    fetch = data.InputFetcher(loader)
    seen = [next(fetch) for _ in range(5)]
    assert all(s.img_exp.shape == (3, 3, 16, 16) for s in seen)

    test_loader = data.get_test_loader(str(tmp_path), img_size=24, batch_size=4, num_workers=2)
    batches = list(test_loader)
    assert [b.img_exp.shape[0] for b in batches] == [4, 3]             # no drop_last, listing order
    k = 0
    for batch in batches:
        for j in range(batch.img_exp.shape[0]):
            a, b, name = ds[k]
            assert batch.img_name[j] == name
            assert torch.equal(batch.img_exp[j].cpu(), O.test_transform(arr_of(a), 24))
            assert torch.equal(batch.img_raw[j].cpu(), O.test_transform(arr_of(b), 24))
            k += 1
    # decode in spawned processes (shared-memory segments): the same tensors as the thread workers
    lp = data.get_train_loader(str(tmp_path), img_size=32, resize_size=16, batch_size=3, shuffle=True, num_workers=2,
                               generator=torch.Generator().manual_seed(11), workers="process")
    got_p = list(lp)
    lp.close()
    assert len(got_p) == 2
    for bp, bt in zip(got_p, got):
        assert bp.img_name == bt.img_name and torch.equal(bp.img_exp.cpu(), bt.img_exp.cpu()) and torch.equal(bp.img_raw.cpu(), bt.img_raw.cpu())
    # two ranks sharing one permutation see equally long shards of it (7 images: padded by wrap-around to 8, like DistributedSampler)
    seen = []
    for rank in range(2):
        ld = data.get_train_loader(str(tmp_path), img_size=32, resize_size=16, batch_size=1, num_workers=1, drop_last=False,
                                   generator=torch.Generator().manual_seed(100 + rank), shard=(rank, 2), shard_seed=3)
        seen.append([[b.img_name[0] for b in ld] for _ in range(2)])          # two epochs
        assert len(ld) == len(seen[-1][0])
    for e in range(2):
        perm = torch.randperm(7, generator=torch.Generator().manual_seed(3 + e)).tolist()
        perm = perm + perm[:1]
        assert seen[0][e] == [ds[i][2] for i in perm[0::2]] and seen[1][e] == [ds[i][2] for i in perm[1::2]]
        assert len(seen[0][e]) == len(seen[1][e]) == 4
    # every rank has the same number of batches whatever drop_last (a per-batch all-reduce loop cannot deadlock), and set_epoch
    # re-synchronises the permutation seed
    lens = [len(data.get_train_loader(str(tmp_path), img_size=32, resize_size=16, batch_size=3, num_workers=1, drop_last=True, shard=(r, 2))) for r in range(2)]
    assert lens[0] == lens[1] == 1
    ld = data.get_train_loader(str(tmp_path), img_size=32, resize_size=16, batch_size=1, num_workers=1, drop_last=False,
                               generator=torch.Generator().manual_seed(100), shard=(0, 2), shard_seed=3)
    ld.set_epoch(1)
    assert [b.img_name[0] for b in ld] == seen[0][1]
    it = iter(ld)          # an abandoned iterator: its pending decodes are waited for before the slots are reused
    next(it)
    assert len(list(ld)) == 4
    # the test loop (tester.py:40-105): enhance, write the PNGs, PSNR / SSIM against the labels -- files and numbers agree with the oracle
    from PIL import Image
    from uegan_amd import models, ops, tester
    ops.set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    G = models.Generator(8, "none", "LeakyReLU", False).to(batches[0].img_exp.device)
    res = tester.run_test(G, data.get_test_loader(str(tmp_path), img_size=32, batch_size=4, num_workers=2), save_dir=str(tmp_path / "out"), tag="1.00")
    assert len(res["names"]) == 7 and len(res["psnr"]) == 7 and abs(res["mean_psnr"] - sum(res["psnr"]) / 7) < 1e-12
    a0, b0, n0 = ds[0]
    want = O.to_uint8_image(tester.enhance(G, O.test_transform(arr_of(b0), 32)[None].to(batches[0].img_exp.device)).cpu())[0].numpy()
    got_png = np.asarray(Image.open(tmp_path / "out" / ("%s_1.00_testFakeExp.png" % n0)))
    assert np.array_equal(got_png, want)
    assert abs(res["psnr"][0] - O.psnr_u8(want, O.to_uint8_image(O.test_transform(arr_of(a0), 32)[None])[0].numpy())) < 1e-9
    # a crop larger than an image is the reference's error
    with pytest.raises(ValueError):
        list(data.get_train_loader(str(tmp_path), img_size=50, resize_size=16, batch_size=2, num_workers=1))
