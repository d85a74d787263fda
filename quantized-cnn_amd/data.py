"""CIFAR-10 input pipeline with the dataset resident in HBM.

The reference reads the python-version CIFAR-10 batches, keeps them as one (N, 32, 32, 3) uint8 array
and builds every image on DataLoader worker processes with PIL + torchvision transforms
(adp_qtz_fan.py:164-193, 452-530).  Here the uint8 array is uploaded once (146 MB for the training
split) and each batch is produced by one kernel launch (`ops.augment_batch`): gather, random crop
with zero padding, random mirror, ToTensor, Normalize.

    data, label = load_data(data_dir, "train")                  # same split as the reference
    loader = GpuLoader(data, label, batch_size=128, train=True)
    for x, y in loader: ...                                      # x (B,3,32,32) fp32 cuda, y (B,) int64 cuda

The batch files are pickles; they are read with an unpickler that only admits the types those files
consist of (dict / list / bytes / int and numpy arrays), so loading executes nothing else.
"""
import io
import os
import pickle

import numpy as np
import torch

from . import ops

MEAN = (0.4915, 0.4821, 0.4462)   # adp_qtz_fan.py:168
STD = (0.2472, 0.2437, 0.2617)
PAD = 4                            # RandomCrop(32, padding=4), adp_qtz_fan.py:165


class _ArrayOnlyUnpickler(pickle.Unpickler):
    _ALLOWED = {
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy", "ndarray"), ("numpy", "dtype"),
        ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    }

    def find_class(self, module, name):
        if (module, name) not in self._ALLOWED:
            raise pickle.UnpicklingError("CIFAR batch file refers to %s.%s, which a batch file has no business with" % (module, name))
        return super().find_class(module, name)


def unpickle(file_name):
    """adp_qtz_fan.py:452-455 (`pickle.load(fo, encoding='bytes')`), restricted to plain data."""
    with open(file_name, "rb") as fo:
        return _ArrayOnlyUnpickler(io.BytesIO(fo.read()), encoding="bytes").load()


def load_data(data_dir, data_type):
    """adp_qtz_fan.py:458-507: "train" = data_batch_1..4 + first half of data_batch_5, "valid" = second
    half of data_batch_5, "test" = test_batch.  Returns (data uint8 (N, 32, 32, 3), label int64 (N,))."""
    def batch(name):
        d = unpickle(os.path.join(data_dir, name))
        return np.array(d[b"data"]), np.array(d[b"labels"])

    if data_type == "train":
        parts = [batch("data_batch_%d" % (i + 1)) for i in range(4)]
        d5, l5 = batch("data_batch_5")
        half = int(len(d5) / 2)
        parts.append((d5[0:half], l5[0:half]))
        data = np.concatenate([p[0] for p in parts])
        label = np.concatenate([p[1] for p in parts])
    elif data_type == "valid":
        d5, l5 = batch("data_batch_5")
        half = int(len(d5) / 2)
        data, label = d5[half:], l5[half:]
    elif data_type == "test":
        data, label = batch("test_batch")
    else:
        raise ValueError("Wrong data type {}".format(data_type))
    data = np.transpose(np.reshape(data, (-1, 3, 32, 32)), (0, 2, 3, 1))
    return np.ascontiguousarray(data, dtype=np.uint8), np.asarray(label, dtype=np.int64)


class GpuLoader:
    """Iterable over batches like the reference's DataLoader(shuffle=train, drop_last=False) with
    transform_train (train=True) or transform_valid (train=False), adp_qtz_fan.py:164-193.  All random
    draws come from one torch.Generator on the device (seedable; statistically the reference's
    transforms, not its random stream)."""

    def __init__(self, data, label, batch_size, train, device="cuda", seed=None, mean=MEAN, std=STD, pad=PAD):
        data = torch.as_tensor(data)
        if data.dim() != 4 or data.dtype != torch.uint8:
            raise ValueError("GpuLoader: data must be uint8 (N, H, W, C), got %s %s" % (data.dtype, tuple(data.shape)))
        self.data = data.to(device).contiguous()
        self.label = torch.as_tensor(label, dtype=torch.int64).to(device)
        if self.label.numel() != self.data.shape[0]:
            raise ValueError("GpuLoader: %d labels for %d images" % (self.label.numel(), self.data.shape[0]))
        self.batch_size, self.train = int(batch_size), bool(train)
        self.mean, self.std, self.pad = mean, std, int(pad)
        self.gen = torch.Generator(device=self.data.device)
        if seed is not None:
            self.gen.manual_seed(int(seed))

    def __len__(self):
        return (self.data.shape[0] + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n, dev = self.data.shape[0], self.data.device
        order = torch.randperm(n, device=dev, generator=self.gen).to(torch.int32) if self.train else torch.arange(n, device=dev, dtype=torch.int32)
        for i in range(0, n, self.batch_size):
            idx = order[i:i + self.batch_size].contiguous()
            b = idx.numel()
            if self.train:
                off = torch.randint(0, 2 * self.pad + 1, (2, b), device=dev, generator=self.gen, dtype=torch.int32)
                flip = (torch.rand(b, device=dev, generator=self.gen) < 0.5).to(torch.uint8)
                x = ops.augment_batch(self.data, idx, off[0].contiguous(), off[1].contiguous(), flip, self.pad, self.mean, self.std)
            else:
                x = ops.augment_batch(self.data, idx, None, None, None, self.pad, self.mean, self.std)
            yield x, self.label[idx.long()]
