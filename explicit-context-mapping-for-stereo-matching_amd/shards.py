"""Packed frame shards: file format, writer from the reference's per-frame `.npy` files, memory-mapped
reader, and a feeder that keeps a GPU supplied one step ahead.

The reference stores every stereo frame as one float32 `.npy` of shape [H,W,7] -- left RGB, right RGB, disparity -- written
by its converters (flying3ddata.py:34-39: two `cv2.imread` uint8 images and a PFM disparity concatenated, which numpy
promotes to float32: 28 bytes per pixel) and read back with `np.load` per sample in `Flying3d.__getitem__` /
`KITTI.__getitem__` (cmf/loader/Flying3d.py:49-80, cmf/loader/KITTI.py:73-108).  The colour channels are integers 0..255,
so a shard keeps them as uint8 and the disparity as fp16 (8 bytes per pixel; fp16 rounds a disparity below 256 px by at
most 2^-4 px) or fp32 (10 bytes per pixel, lossless).  `ecm_frame_prep_packed` (csrc/frame_prep.hip) decodes a batch of
packed frames on the device into exactly what the reference's loader + transform return.

File layout (little endian):

    [0, 64)        header: magic "ECMSHRD1", u32 version (1), u32 n_frames, u32 disparity bytes (2 | 4), u32 flags (0),
                   u64 index_offset, u64 names_offset, u64 data_offset, u64 file_bytes, zero padding
    index_offset   n_frames x 48 bytes: u64 rgb_offset, u64 disp_offset, u32 H, u32 W, i32 hmin, hmax, wmin, wmax
                   (bounding box of the non-zero disparities: KITTI's training crop draws its window from it,
                   KITTI.py:84-94), 8 bytes padding
    names_offset   n_frames x 64 bytes: source file name, utf-8, zero padded (order = the order the reference's dataset
                   object lists them in: sorted for Flying3d, Flying3d.py:31-32)
    data_offset    (4096-aligned) per frame: rgb6 uint8 [H,W,6] then disparity [H,W], each 64-byte aligned

Everything here is host-side plumbing in Python, like the reference's loaders; the decode is the HIP kernel.
"""
from __future__ import annotations

import os
import queue
import random
import struct
import threading

import numpy as np
import torch

MAGIC = b"ECMSHRD1"
VERSION = 1
_HEADER = struct.Struct("<8sIIIIQQQQ")          # 56 bytes used of 64
_HEADER_BYTES = 64
_INDEX = np.dtype([("rgb_off", "<u8"), ("disp_off", "<u8"), ("H", "<u4"), ("W", "<u4"), ("hmin", "<i4"), ("hmax", "<i4"),
                   ("wmin", "<i4"), ("wmax", "<i4"), ("pad", "<u8")])
assert _INDEX.itemsize == 48
_NAME_BYTES = 64


def _align(n, a):
    return (n + a - 1) // a * a


# ------------------------------------------------------------------------------------------------------ writer
def list_frames(directory):
    """The reference's file order: `os.listdir` then `sort()` (Flying3d.py:31-32) -- lexicographic, so "10.npy" < "2.npy"."""
    files = [f for f in os.listdir(directory) if f.endswith(".npy")]
    files.sort()
    if len(files) < 1:
        raise Exception("No files found in %s" % directory)          # Flying3d.py:34-35
    return [os.path.join(directory, f) for f in files]


def write_shard(frames, out_path, disparity="fp16", names=None):
    """frames: iterable of [H,W,7] arrays (or `.npy` paths as written by the reference's converters, flying3ddata.py:41)
    -> one shard file.  Colour channels must hold integers 0..255 (they come from uint8 images); anything else is refused,
    never rounded silently.  disparity: "fp16" (8 B/px) or "fp32" (10 B/px, lossless)."""
    if disparity not in ("fp16", "fp32"):
        raise ValueError("disparity must be 'fp16' or 'fp32'")
    dbytes, dtype = (2, np.float16) if disparity == "fp16" else (4, np.float32)
    frames = list(frames)
    n = len(frames)
    if n < 1:
        raise Exception("No frames to write")
    if names is None:
        names = [os.path.basename(f) if isinstance(f, (str, os.PathLike)) else f"{i}.npy" for i, f in enumerate(frames)]
    index = np.zeros(n, _INDEX)
    index_off = _HEADER_BYTES
    names_off = index_off + n * _INDEX.itemsize
    data_off = _align(names_off + n * _NAME_BYTES, 4096)
    tmp = out_path + ".tmp"
    with open(tmp, "wb") as f:
        f.seek(data_off)
        pos = data_off
        for i, fr in enumerate(frames):
            a = np.load(fr, mmap_mode="r") if isinstance(fr, (str, os.PathLike)) else np.asarray(fr)
            if a.ndim != 3 or a.shape[2] != 7:
                raise ValueError(f"frame {names[i]}: expected [H,W,7] (left RGB, right RGB, disparity), got {a.shape}")
            H, W = a.shape[:2]
            col = np.asarray(a[..., :6])
            rgb = col.astype(np.uint8)
            if not np.array_equal(rgb.astype(col.dtype), col):
                raise ValueError(f"frame {names[i]}: colour channels are not integers in 0..255; a packed shard cannot hold them")
            dsp = np.asarray(a[..., 6], dtype=np.float32)
            nz = np.nonzero(dsp)
            box = (int(nz[0].min()), int(nz[0].max()), int(nz[1].min()), int(nz[1].max())) if nz[0].size else (0, H - 1, 0, W - 1)
            rgb_off = _align(pos, 64)
            disp_off = _align(rgb_off + H * W * 6, 64)
            f.seek(rgb_off)
            f.write(np.ascontiguousarray(rgb).tobytes())
            f.seek(disp_off)
            f.write(np.ascontiguousarray(dsp.astype(dtype)).tobytes())
            pos = disp_off + H * W * dbytes
            index[i] = (rgb_off, disp_off, H, W, box[0], box[1], box[2], box[3], 0)
        total = _align(pos, 4096)
        f.truncate(total)
        f.seek(0)
        f.write(_HEADER.pack(MAGIC, VERSION, n, dbytes, 0, index_off, names_off, data_off, total).ljust(_HEADER_BYTES, b"\0"))
        f.seek(index_off)
        f.write(index.tobytes())
        f.seek(names_off)
        for nm in names:
            b = str(nm).encode("utf-8")[:_NAME_BYTES]
            f.write(b.ljust(_NAME_BYTES, b"\0"))
    os.replace(tmp, out_path)
    return out_path


def write_shard_from_directory(directory, out_path, disparity="fp16"):
    """One shard from a directory of the reference's frames (e.g. <flying3d root>/train), in the reference's file order."""
    return write_shard(list_frames(directory), out_path, disparity)


# ------------------------------------------------------------------------------------------------------ reader
class ShardReader:
    """Memory-mapped view of a shard: `frame(i)` -> (rgb6 uint8 [H,W,6], disparity fp16|fp32 [H,W]) numpy views into the
    page cache (no copy, no decode)."""

    def __init__(self, path):
        self.path = path
        size = os.path.getsize(path)
        if size < _HEADER_BYTES:
            raise ValueError(f"{path}: too short to be a shard")
        self._map = np.memmap(path, dtype=np.uint8, mode="r")
        magic, version, n, dbytes, flags, index_off, names_off, data_off, total = _HEADER.unpack(bytes(self._map[:_HEADER.size]))
        if magic != MAGIC:
            raise ValueError(f"{path}: not a shard (magic {magic!r})")
        if version != VERSION:
            raise ValueError(f"{path}: shard version {version}, this reader handles {VERSION}")
        if dbytes not in (2, 4) or total != size or n < 1:
            raise ValueError(f"{path}: corrupt header (disparity bytes {dbytes}, {n} frames, {total} bytes recorded, {size} on disk)")
        self.n, self.disp_bytes = int(n), int(dbytes)
        self.disp_dtype = np.float16 if dbytes == 2 else np.float32
        self.index = np.frombuffer(self._map, dtype=_INDEX, count=self.n, offset=int(index_off))
        raw = bytes(self._map[int(names_off):int(names_off) + self.n * _NAME_BYTES])
        self.names = [raw[i * _NAME_BYTES:(i + 1) * _NAME_BYTES].rstrip(b"\0").decode("utf-8") for i in range(self.n)]
        last = self.index[-1]
        if int(last["disp_off"]) + int(last["H"]) * int(last["W"]) * self.disp_bytes > size:
            raise ValueError(f"{path}: frame table points past the end of the file")

    def __len__(self):
        return self.n

    def shape(self, i):
        e = self.index[i]
        return int(e["H"]), int(e["W"])

    def valid_box(self, i):
        e = self.index[i]
        return int(e["hmin"]), int(e["hmax"]), int(e["wmin"]), int(e["wmax"])

    def frame(self, i):
        e = self.index[i]
        H, W = int(e["H"]), int(e["W"])
        ro, do = int(e["rgb_off"]), int(e["disp_off"])
        rgb = self._map[ro:ro + H * W * 6].reshape(H, W, 6)
        dsp = self._map[do:do + H * W * self.disp_bytes].view(self.disp_dtype).reshape(H, W)
        return rgb, dsp

    def frame_float32(self, i):
        """The [H,W,7] float32 array the reference's `np.load` would return for this frame (disparity as stored)."""
        rgb, dsp = self.frame(i)
        return np.concatenate([rgb.astype(np.float32), dsp.astype(np.float32)[..., None]], 2)


# ------------------------------------------------------------------------------------------------------ feeder
class ShardFeeder:
    """Iterates batches of (left, right, disparity) device tensors -- what the reference's DataLoader over Flying3d / KITTI
    yields (train.py:154-161) -- from a shard, one or more steps ahead of the consumer:

        host thread:   gather the batch's frames (train: only the crop window) from the memory map into a pinned buffer
        copy stream:   H2D of the packed bytes (8 or 10 B/px), then ecm_frame_prep_packed on the same stream
        consumer:      `next()` makes the compute stream wait for that batch's event; no host synchronisation

    split: "train" (random 256x512 windows, Flying3d.py:51-56; with `kitti=True` the window is drawn inside the bounding
    box of the valid disparities, KITTI.py:84-97), "test" (Flying3d's 540 -> 576 tail padding, Flying3d.py:66-72) or
    "kitti_test" (top / left padding to 384x1248, KITTI.py:98-108; batch 1 or equal-sized frames).
    Data parallel: rank r of `world` takes every world-th batch of the epoch's order (no two ranks see the same frame in an
    epoch).  The order is a seeded permutation per epoch (the reference shuffles with the DataLoader's default generator)."""

    def __init__(self, reader, batch, split="train", device="cuda", rank=0, world=1, seed=0, prefetch=2, shuffle=None,
                 kitti=False, crop=(256, 512), want_image=False):
        from . import ops
        self.ops = ops
        self.reader, self.batch, self.split, self.kitti = reader, int(batch), split, bool(kitti)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ShardFeeder feeds the MI355X HIP path: a CUDA/HIP device is required (no CPU fallback exists)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.rank, self.world, self.seed = int(rank), int(world), int(seed)
        self.shuffle = (split == "train") if shuffle is None else bool(shuffle)
        self.crop, self.want_image = crop, want_image
        self.prefetch = max(1, int(prefetch))
        self._copy = torch.cuda.Stream(device=self.device)
        self._slots = None
        self._q = None
        self._thread = None
        self._stop = threading.Event()
        self._err = None
        self.epoch = 0

    # -- sampling ------------------------------------------------------------------------------------------------------
    def batches_per_epoch(self):
        return (len(self.reader) // self.batch) // self.world

    def _order(self, epoch):
        n = len(self.reader)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed * 1000003 + epoch)
            perm = torch.randperm(n, generator=g).tolist()
        else:
            perm = list(range(n))
        nb = (n // self.batch) // self.world * self.world                   # whole batches, equal count per rank
        return [perm[b * self.batch:(b + 1) * self.batch] for b in range(self.rank, nb, self.world)]

    def _window(self, i, rng):
        H, W = self.reader.shape(i)
        th, tw = self.crop
        if self.kitti:                                                      # KITTI.py:84-97
            hmin, hmax, wmin, wmax = self.reader.valid_box(i)
            if hmax - hmin <= th:
                hmin = hmax - th
            if wmax - wmin <= tw:
                wmax = wmin + tw
            return rng.randint(hmin, hmax - th), rng.randint(wmin, wmax - tw)
        return rng.randint(0, H - th), rng.randint(0, W - tw)              # Flying3d.py:54-55

    # -- pipeline ------------------------------------------------------------------------------------------------------
    def _make_slot(self, H, W):
        pin = dict(rgb=torch.empty(self.batch, H, W, 6, dtype=torch.uint8).pin_memory(),
                   dsp=torch.empty(self.batch, H, W, dtype=torch.float16 if self.reader.disp_bytes == 2 else torch.float32).pin_memory())
        dev = dict(rgb=torch.empty(self.batch, H, W, 6, dtype=torch.uint8, device=self.device),
                   dsp=torch.empty(self.batch, H, W, dtype=pin["dsp"].dtype, device=self.device))
        return dict(pin=pin, dev=dev, shape=(H, W), h2d_done=None)

    def _produce(self, epoch):
        try:                                                                # whatever goes wrong reaches the consumer
            torch.cuda.set_device(self.device)
            rng = random.Random(self.seed * 7919 + epoch * 104729 + self.rank)
            k = 0
            for idx in self._order(epoch):
                if self._stop.is_set():
                    break
                if self.split == "train":
                    H, W = self.crop
                    wins = [self._window(i, rng) for i in idx]
                else:
                    shapes = {self.reader.shape(i) for i in idx}
                    if len(shapes) != 1:
                        raise RuntimeError(f"frames of different sizes in one batch {sorted(shapes)}: use batch 1")
                    (H, W), wins = next(iter(shapes)), None
                slot = self._slots[k % len(self._slots)]
                if slot is None or slot["shape"] != (H, W):
                    slot = self._slots[k % len(self._slots)] = self._make_slot(H, W)
                if slot["h2d_done"] is not None:
                    slot["h2d_done"].synchronize()                          # the previous copy out of this pinned buffer
                rgb_np, dsp_np = slot["pin"]["rgb"].numpy(), slot["pin"]["dsp"].numpy()
                for b, i in enumerate(idx):
                    rgb, dsp = self.reader.frame(i)
                    if wins is not None:
                        y0, x0 = wins[b]
                        rgb, dsp = rgb[y0:y0 + H, x0:x0 + W], dsp[y0:y0 + H, x0:x0 + W]
                    np.copyto(rgb_np[b], rgb)
                    np.copyto(dsp_np[b], dsp)
                with torch.cuda.stream(self._copy):
                    slot["dev"]["rgb"].copy_(slot["pin"]["rgb"], non_blocking=True)
                    slot["dev"]["dsp"].copy_(slot["pin"]["dsp"], non_blocking=True)
                    slot["h2d_done"] = torch.cuda.Event()
                    slot["h2d_done"].record(self._copy)
                    frames = (slot["dev"]["rgb"], slot["dev"]["dsp"])
                    if self.split == "train":
                        out = self.ops.frame_prep(frames, [0] * self.batch, [0] * self.batch, H, W, want_image=self.want_image)
                    elif self.split == "test":                             # 540 rows + the last 36 again (Flying3d.py:66-72)
                        out = self.ops.frame_prep(frames, [0] * self.batch, [0] * self.batch, min(H, 540) + 36, min(W, 960),
                                                  split=min(H, 540), tail=36, want_image=self.want_image)
                    elif self.split == "kitti_test":
                        out = self.ops.frame_prep_kitti_eval(frames, want_image=self.want_image)
                    else:
                        raise ValueError(f"unknown split {self.split!r}")
                    ready = torch.cuda.Event()
                    ready.record(self._copy)
                self._q.put((out, ready, idx, wins))
                k += 1
            self._q.put(None)
        except BaseException as e:                                          # surfaced by the consumer
            self._err = e
            self._q.put(None)

    def __iter__(self):
        self.close()
        self._stop.clear()
        self._err = None
        self._slots = [None] * (self.prefetch + 1)
        self._q = queue.Queue(maxsize=self.prefetch)
        self._thread = threading.Thread(target=self._produce, args=(self.epoch,), daemon=True)
        self._thread.start()
        self.epoch += 1
        return self

    def __next__(self):
        while True:
            try:
                item = self._q.get(timeout=1.0)
                break
            except queue.Empty:                                             # a producer that died without a word
                if not self._thread.is_alive() and self._q.empty():
                    raise RuntimeError("ShardFeeder: the producer thread ended without delivering a batch") from self._err
        if item is None:
            self._thread.join()
            if self._err is not None:
                raise self._err
            raise StopIteration
        out, ready, idx, wins = item
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ready)
        for t in out:
            if t is not None:
                t.record_stream(cur)
        self.last_indices, self.last_windows = idx, wins
        return out

    def close(self):
        if self._thread is not None and self._thread.is_alive():
            self._stop.set()
            try:
                while self._q.get(timeout=5.0) is not None:
                    pass
            except queue.Empty:
                pass
            self._thread.join(timeout=10.0)
        self._thread = None
