"""Input pipeline with the transform ON THE DEVICE: the reference's data_loader.py restated so that the host
only decodes files and copies raw bytes, and everything torchvision does per pixel runs in uegan_input_transform.

  reference (data_loader.py)                               here
  ---------------------------------------------------     ------------------------------------------------------------------
  ReferenceDataset / DefaultDataset  (:21-71)               same names: file listing + pairing only (no transform argument)
  transforms.RandomCrop(img_size)    (:75)                  the crop window is what gets copied: host memcpy of its rows into a
                                                            pinned batch buffer, one H2D copy per batch on a side stream
  Resize / flips / ToTensor / Normalize (:76-81, :97-100)   uegan_input_transform (csrc/input.hip), bit-exact with Pillow's resampler
  DataLoader(pin_memory=True) + InputFetcher (:85-90,       DeviceLoader: decode threads -> pinned ring -> side-stream H2D + transform,
      :113-133: `.to(device)` on the training stream)       `prefetch` batches ahead; the training stream only waits on an event

Random draws follow torchvision's call order per image (RandomCrop.get_params: randint for the top row, randint for the left
column -- none when the image already has the crop size; then one torch.rand(1) < 0.5 per flip), img_exp before img_raw as
ReferenceDataset.__getitem__ transforms them (:63-65), and torch.randperm for the shuffle.  With DataLoader worker processes the
reference's own streams are per-worker and not reproducible, so this order is a convention, not a parity claim (torchvision is not
in this image: unpinned); the PIXEL arithmetic is pinned against Pillow itself (tests/test_data.py).
"""
import collections
import concurrent.futures
import math
import os
from pathlib import Path

import numpy as np
import torch

from . import _lib as L
from .ops import _p, lib

_EXTS = ("png", "jpg", "jpeg", "JPG")
PRECISION_BITS = 32 - 8 - 2           # Pillow, src/libImaging/Resample.c (8-bit channels)
MAX_IMAGES_PER_CALL = 64


# --------------------------------------------------------------------------------------------------------------------
# Pillow's coefficient tables (Resample.c: precompute_coeffs + normalize_coeffs_8bpc) for the BILINEAR (triangle) filter
# --------------------------------------------------------------------------------------------------------------------
def _triangle(x):
    x = -x if x < 0.0 else x
    return 1.0 - x if x < 1.0 else 0.0


def resample_table(in_size, out_size):
    """int32 [out_size, 2 + k]: per output index (first input index, tap count, k fixed-point coefficients), and k.
    Python floats are C doubles and the operations are in Pillow's order, so the integers are Pillow's."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale                      # bilinear: filter support 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    tab = np.zeros((out_size, 2 + ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)           # C (int) cast: truncation
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        n = xmax - xmin
        k = [_triangle((x + xmin - center + 0.5) * ss) for x in range(n)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        tab[xx, 0], tab[xx, 1] = xmin, n
        for x, w in enumerate(k):
            tab[xx, 2 + x] = int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS))
    return tab, ksize


_TABLES = {}


def _device_table(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    if key not in _TABLES:
        tab, k = resample_table(in_size, out_size)
        _TABLES[key] = (torch.from_numpy(tab).to(device), k)
    return _TABLES[key]


def input_transform(pixels, out_size, flips=None, out=None, stream=None):
    """pixels: uint8 [B, h, w, 3] on the device (decoded RGB crop windows) -> fp32 [B, 3, out_h, out_w] in [-1, 1]:
    Resize(out_size) + flips (bit 0 horizontal, bit 1 vertical, per image) + ToTensor + Normalize(0.5, 0.5)."""
    if pixels.dtype != torch.uint8 or pixels.dim() != 4 or pixels.shape[3] != 3 or not pixels.is_contiguous():
        raise ValueError("input_transform expects a contiguous uint8 [B, h, w, 3] tensor")
    B, h, w, _ = pixels.shape
    oh, ow = (out_size, out_size) if isinstance(out_size, int) else out_size
    dev = pixels.device
    htab, hk = _device_table(w, ow, dev)
    vtab, vk = _device_table(h, oh, dev)
    if out is None:
        out = torch.empty((B, 3, oh, ow), dtype=torch.float32, device=dev)
    if stream is None and not L.is_emulated():
        stream = torch.cuda.current_stream().cuda_stream
    for b0 in range(0, B, MAX_IMAGES_PER_CALL):
        nb = min(MAX_IMAGES_PER_CALL, B - b0)
        tmp = torch.empty((nb, h, ow, 3), dtype=torch.uint8, device=dev)
        fl = None
        if flips is not None:
            fl = (L.C.c_int32 * nb)(*[int(f) for f in flips[b0:b0 + nb]])
        L.check(lib().uegan_input_transform(_p(pixels[b0:]), nb, h, w, oh, ow, _p(htab), hk, _p(vtab), vk, fl, _p(tmp), _p(out[b0:]), stream))
    return out


# --------------------------------------------------------------------------------------------------------------------
# datasets: file listing only
# --------------------------------------------------------------------------------------------------------------------
def listdir(dname):
    """every image file below dname (data_loader.py:14-17: recursive, extension groups in this order)"""
    out = []
    for ext in _EXTS:
        out.extend(Path(dname).rglob("*." + ext))
    return out


class DefaultDataset:
    """sorted image files of one folder (data_loader.py:21-36)"""

    def __init__(self, root):
        self.samples = sorted(listdir(root))

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, index):
        return self.samples[index]


class ReferenceDataset:
    """pairs (file of the first sub-folder, file of the second) in listing order (data_loader.py:39-71); the sample name is the
    second file's path up to its first '.', after its last '/' (:58-60)"""

    def __init__(self, root):
        firsts, seconds = [], []
        for idx, domain in enumerate(sorted(os.listdir(root))):
            files = listdir(os.path.join(root, domain))
            if idx == 0:
                firsts += files
            elif idx == 1:
                seconds += files
        self.samples = list(zip(firsts, seconds))

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, index):
        a, b = self.samples[index]
        stem = str(b).split(".", 1)[0]
        return a, b, stem.rsplit("/", 1)[1]


# --------------------------------------------------------------------------------------------------------------------
# the loader
# --------------------------------------------------------------------------------------------------------------------
def draw_train_params(h, w, crop, generator=None):
    """(top, left, flip bits) in torchvision's draw order for RandomCrop -> RandomHorizontalFlip -> RandomVerticalFlip"""
    if h < crop or w < crop:
        raise ValueError("Required crop size (%d, %d) is larger than input image size (%d, %d)" % (crop, crop, h, w))
    if h == crop and w == crop:
        top = left = 0
    else:
        top = int(torch.randint(0, h - crop + 1, size=(1,), generator=generator).item())
        left = int(torch.randint(0, w - crop + 1, size=(1,), generator=generator).item())
    bits = 0
    if float(torch.rand(1, generator=generator)) < 0.5:
        bits |= 1
    if float(torch.rand(1, generator=generator)) < 0.5:
        bits |= 2
    return top, left, bits


Batch = collections.namedtuple("Batch", ["img_exp", "img_raw", "img_name"])


class _Slot:
    """one batch in flight: a pinned byte buffer (thread workers) or a shared-memory segment registered with the HIP runtime
    (process workers), its device mirror, the decode futures and the 'ready' event"""

    def __init__(self):
        self.host = None
        self.dev = None
        self.shm = None
        self.registered = False
        self.futures = []
        self.items = None
        self.ready = None
        self.out = None
        self.copied = None

    def ensure(self, nbytes, device, pinned, shared):
        if self.host is not None and self.host.numel() >= nbytes:
            return
        if shared:
            from multiprocessing import shared_memory
            self.release()
            self.shm = shared_memory.SharedMemory(create=True, size=max(nbytes, 1 << 20))
            self.host = torch.frombuffer(self.shm.buf, dtype=torch.uint8)
            if pinned:       # page-lock the segment so that the H2D copy is a DMA straight out of what the workers wrote
                self.registered = int(torch.cuda.cudart().cudaHostRegister(self.host.data_ptr(), self.host.numel(), 0)) == 0
        else:
            self.host = torch.empty((nbytes,), dtype=torch.uint8, pin_memory=pinned)
        self.dev = torch.empty((self.host.numel(),), dtype=torch.uint8, device=device)

    def release(self):
        if self.shm is not None:
            if self.registered:
                torch.cuda.cudart().cudaHostUnregister(self.host.data_ptr())
                self.registered = False
            self.host = None
            try:
                self.shm.close()
                self.shm.unlink()
            except (BufferError, FileNotFoundError):
                pass
            self.shm = None


_ATTACHED = {}


def _shm_view(name, off, h, w):
    """worker process: numpy view of one window inside the named shared-memory segment (segments stay attached per process)"""
    from multiprocessing import shared_memory
    if name not in _ATTACHED:
        if len(_ATTACHED) > 16:
            for old in list(_ATTACHED.values()):
                old.close()
            _ATTACHED.clear()
        _ATTACHED[name] = shared_memory.SharedMemory(name=name)
    return np.ndarray((h, w, 3), dtype=np.uint8, buffer=_ATTACHED[name].buf, offset=off)


def _worker_decode(name, off, path, top, left, h, w, whole):
    dst = _shm_view(name, off, h, w)
    if whole:
        _decode_whole(path, dst)
    else:
        _decode_window(path, top, left, h, w, dst)
    return None


def _decode_window(path, top, left, h, w, dst):
    from PIL import Image
    with Image.open(path) as im:
        im.load()                                            # the decoder runs without the GIL
        win = im.crop((left, top, left + w, top + h))        # only the window goes through the mode conversion and the copies
        if win.mode != "RGB":
            win = win.convert("RGB")
        dst[...] = np.asarray(win)


def _decode_whole(path, dst):
    from PIL import Image
    with Image.open(path) as im:
        dst[...] = np.asarray(im.convert("RGB"))


_SIZES = {}


def _image_size(path):
    """(h, w) from the file header, remembered per path (the files of a dataset come round every epoch)"""
    key = str(path)
    if key not in _SIZES:
        from PIL import Image
        with Image.open(path) as im:          # header only
            _SIZES[key] = (im.size[1], im.size[0])
    return _SIZES[key]


class DeviceLoader:
    """Iterates Batch(img_exp, img_raw, img_name): fp32 [B,3,S,S] tensors on `device`, exactly the tensors the reference's
    DataLoader + InputFetcher deliver (data_loader.py:113-133), produced `prefetch` batches ahead of the consumer.

    train=True : RandomCrop(img_size) -> Resize(resize_size) -> flips      (get_train_loader, :72-90)
    train=False: Resize(img_size) of the whole image                        (get_test_loader, :93-110)"""

    def __init__(self, dataset, batch_size, img_size=512, resize_size=256, train=True, shuffle=True, drop_last=True, num_workers=8,
                 device=None, prefetch=2, generator=None, shard=None, shard_seed=0, workers="thread"):
        """shard = (rank, world_size): this process iterates samples rank, rank + world, ... of the (shuffled) order -- one
        loader per GPU process.  The permutation of epoch e is then drawn from its own generator seeded
        shard_seed + e (the same on every rank, like DistributedSampler.set_epoch); crops and flips stay per-rank draws.
        workers: "thread" (decode in threads of this process: enough up to ~600 img/s, then the interpreter lock shared with the
        training loop caps it) or "process" (spawned decode processes writing into shared-memory segments that are page-locked
        for the H2D copy -- what DataLoader(num_workers=N) does, minus the transform and the pickling of tensors; as with any
        spawned pool, the launching script needs its `if __name__ == "__main__":` guard)."""
        self.dataset, self.batch_size, self.img_size, self.resize_size = dataset, batch_size, img_size, resize_size
        self.train, self.shuffle, self.drop_last = train, shuffle, drop_last
        self.generator = generator
        self.shard, self.shard_seed, self.epoch = shard, shard_seed, 0
        self.emulated = L.is_emulated()
        self.device = torch.device("cpu") if self.emulated else torch.device(device if device is not None else "cuda")
        self.prefetch = max(1, prefetch)
        if workers not in ("thread", "process"):
            raise ValueError("workers must be 'thread' or 'process'")
        self.shared = workers == "process"
        if self.shared:
            import multiprocessing
            self.pool = concurrent.futures.ProcessPoolExecutor(max_workers=max(1, num_workers), mp_context=multiprocessing.get_context("spawn"))
        else:
            self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, num_workers))
        self.side = None if self.emulated else torch.cuda.Stream(device=self.device)
        self.slots = [_Slot() for _ in range(self.prefetch + 1)]

    def _n(self):
        n = len(self.dataset)
        if self.shard is not None:            # every rank's shard has the same length (DistributedSampler's padding): see _order
            world = self.shard[1]
            n = (n + world - 1) // world
        return n

    def set_epoch(self, epoch):
        """the permutation of epoch e is seeded with shard_seed + e on every rank (DistributedSampler.set_epoch).  Without a call the
        loader counts its own iterations -- which stays in step across ranks because every rank's shard has the same number of batches."""
        self.epoch = int(epoch)

    def close(self):
        """stop the decode workers and free the shared-memory segments"""
        self.pool.shutdown(wait=True, cancel_futures=True)
        for sl in self.slots:
            if sl.copied is not None:
                sl.copied.synchronize()
            sl.release()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        n = self._n()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _order(self):
        n = len(self.dataset)
        if self.shard is None:
            return torch.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))
        rank, world = self.shard
        g = torch.Generator().manual_seed(self.shard_seed + self.epoch)
        self.epoch += 1
        order = torch.randperm(n, generator=g).tolist() if self.shuffle else list(range(n))
        # equal shards: pad by wrap-around to a multiple of `world` (like DistributedSampler with drop_last=False).  Unequal shards
        # would give the ranks different batch counts -- a `for batch in loader: train_step(...)` loop then deadlocks in the gradient
        # all-reduce, and with InputFetcher the short rank restarts early and its epoch counter (the permutation seed) drifts.
        total = (n + world - 1) // world * world
        order = (order * (total // max(n, 1) + 1))[:total] if n else order
        return order[rank::world]

    # -- stage 1: host decode into the pinned buffer (threads) -------------------------------------------------------
    def _submit(self, slot, indices):
        for f in getattr(slot, "futures", None) or []:      # decodes left pending by an abandoned iterator still write into this slot
            try:
                f.result()
            except Exception:
                pass
        slot.futures = []
        if slot.copied is not None:
            slot.copied.synchronize()            # the previous H2D copy out of this pinned buffer has finished
            slot.copied = None
        items = [self.dataset[i] for i in indices]
        plan, off = [], 0
        for a, b, name in items:
            for path in (a, b):
                h, w = _image_size(path)
                if self.train:
                    top, left, bits = draw_train_params(h, w, self.img_size, self.generator)
                    plan.append((path, top, left, self.img_size, self.img_size, bits, off))
                    off += self.img_size * self.img_size * 3
                else:
                    plan.append((path, 0, 0, h, w, 0, off))
                    off += h * w * 3
        slot.ensure(off, self.device, pinned=not self.emulated, shared=self.shared)
        host = slot.host.numpy()
        slot.futures = []
        for path, top, left, h, w, bits, o in plan:
            if self.shared:
                slot.futures.append(self.pool.submit(_worker_decode, slot.shm.name, o, str(path), top, left, h, w, not self.train))
                continue
            dst = host[o:o + h * w * 3].reshape(h, w, 3)
            if self.train:
                slot.futures.append(self.pool.submit(_decode_window, path, top, left, h, w, dst))
            else:
                slot.futures.append(self.pool.submit(_decode_whole, path, dst))
        slot.items, slot.plan, slot.nbytes, slot.out = items, plan, off, None

    # -- stage 2: one H2D copy + the transform kernels, on the side stream -------------------------------------------
    def _upload(self, slot):
        for f in slot.futures:
            f.result()                           # (re-raises decode errors here)
        B = len(slot.items)
        S = self.resize_size if self.train else self.img_size

        def run(stream_handle):
            slot.dev[:slot.nbytes].copy_(slot.host[:slot.nbytes], non_blocking=True)
            out = torch.empty((2 * B, 3, S, S), dtype=torch.float32, device=self.device)
            if self.train:
                c = self.img_size
                pix = slot.dev[:slot.nbytes].view(2 * B, c, c, 3)
                input_transform(pix, S, [p[5] for p in slot.plan], out=out, stream=stream_handle)
            else:
                for i, (path, _, _, h, w, _, o) in enumerate(slot.plan):
                    input_transform(slot.dev[o:o + h * w * 3].view(1, h, w, 3), S, None, out=out[i:i + 1], stream=stream_handle)
            return out

        if self.emulated:
            slot.out = run(None)
            return
        with torch.cuda.stream(self.side):
            slot.out = run(self.side.cuda_stream)
            slot.copied = torch.cuda.Event()
            slot.copied.record(self.side)
            slot.ready = slot.copied

    def __iter__(self):
        order = self._order()
        bs = self.batch_size
        batches = [order[i:i + bs] for i in range(0, len(order), bs)]
        if self.drop_last and batches and len(batches[-1]) < bs:
            batches.pop()
        queue = collections.deque()
        nxt = 0
        free = collections.deque(self.slots)
        while nxt < len(batches) or queue:
            while nxt < len(batches) and free and len(queue) < self.prefetch + 1:
                slot = free.popleft()
                self._submit(slot, batches[nxt])
                queue.append(slot)
                nxt += 1
            head = queue[0]
            if head.out is None:
                self._upload(head)
            for s in list(queue)[1:]:            # upload later batches whose decode has already finished
                if s.out is None and all(f.done() for f in s.futures):
                    self._upload(s)
            queue.popleft()
            out, names = head.out, [it[2] for it in head.items]
            if not self.emulated:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(head.ready)
                out.record_stream(cur)
            B = len(head.items)
            pair = out.view(B, 2, *out.shape[1:])
            head.out = None
            free.append(head)
            yield Batch(pair[:, 0], pair[:, 1], names)


def get_train_loader(root, img_size=512, resize_size=256, batch_size=8, shuffle=True, num_workers=8, drop_last=True, device=None, generator=None,
                     shard=None, shard_seed=0, workers="thread"):
    """data_loader.py:72-90 with the transform on the device"""
    return DeviceLoader(ReferenceDataset(root), batch_size, img_size, resize_size, True, shuffle, drop_last, num_workers, device, generator=generator,
                        shard=shard, shard_seed=shard_seed, workers=workers)


def get_test_loader(root, img_size=512, batch_size=8, shuffle=False, num_workers=4, device=None, generator=None, workers="thread"):
    """data_loader.py:93-110"""
    return DeviceLoader(ReferenceDataset(root), batch_size, img_size, img_size, False, shuffle, False, num_workers, device, generator=generator,
                        workers=workers)


class InputFetcher:
    """data_loader.py:113-133: endless iteration, restarting the loader when it runs out; the tensors are already on the device"""

    def __init__(self, loader):
        self.loader = loader
        self.iter = None

    def _fetch_refs(self):
        if self.iter is None:
            self.iter = iter(self.loader)
        try:
            return next(self.iter)
        except StopIteration:
            self.iter = iter(self.loader)
            return next(self.iter)

    def __next__(self):
        return self._fetch_refs()
