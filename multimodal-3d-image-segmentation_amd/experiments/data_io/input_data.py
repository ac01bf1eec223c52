"""InputData with the GPU-side sample pipeline (reference experiments/data_io/input_data.py:15-151 and
dataset.py:14-60).

Same constructor and flow methods as the reference.  The difference is where the per-sample work runs:
the reference normalises and augments every sample with numpy / SimpleITK inside DataLoader worker processes;
here the reader's raw arrays are uploaded once and ``x_processing`` / ``ImageTransform`` run as HIP kernels,
so the flows yield batches that are already resident in HBM (``training()`` 's ``.to(device)`` is then a
no-op).  ``num_workers`` reader threads prefetch the raw host arrays of the next batches.
"""
import math
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .dataset import ImageTransform


class _Flow:
    """Re-iterable like a DataLoader: every ``iter()`` starts a new epoch (train_test.py:146,189)."""

    def __init__(self, owner, data_lists, shuffle, transform_kwargs, sharded=False):
        self.owner, self.data_lists, self.shuffle = owner, data_lists, shuffle
        self.sharded = sharded      # train / validation flows only: testing() predicts every sample on the rank that runs it
        self.transform = ImageTransform(**transform_kwargs) if transform_kwargs is not None else None
        self._order_rng = np.random.default_rng(owner.shuffle_seed)

    def __len__(self):
        return self.owner._get_num_batches(self.data_lists, self.sharded)

    def _epoch_order(self):
        """Sample order of one epoch for THIS process: the (shared-seed) permutation, sharded rank::world with the uneven
        tail dropped so that every rank runs the same number of batches."""
        o = self.owner
        n = len(self.data_lists[0])
        order = self._order_rng.permutation(n) if self.shuffle else np.arange(n)
        if self.sharded and o.world > 1:
            order = order[:(n // o.world) * o.world][o.rank::o.world]
        return order

    def _read(self, idx):
        o = self.owner
        x = np.stack([np.asarray(o.reader(self.data_lists[m][idx])) for m in o.idx_x_modalities])
        y = None
        if o.idx_y_modalities is not None:
            y = np.stack([np.asarray(o.reader(self.data_lists[m][idx])) for m in o.idx_y_modalities])
        return x, y

    def _sample(self, raw):
        o = self.owner
        x = torch.as_tensor(np.ascontiguousarray(raw[0], dtype=np.float32)).to(o.device, non_blocking=True)
        if o.x_processing is not None:
            x = o.x_processing(x)
        if raw[1] is None:
            return (self.transform(x) if self.transform is not None else x), None
        y = torch.as_tensor(np.ascontiguousarray(raw[1], dtype=np.float32)).to(o.device, non_blocking=True)
        if self.transform is not None:
            x, y = self.transform(x, y)
        return x, y

    def __iter__(self):
        o = self.owner
        order = self._epoch_order()
        batches = [order[i:i + o.batch_size] for i in range(0, len(order), o.batch_size)]
        workers = max(1, int(o.num_workers))
        with ThreadPoolExecutor(workers) as pool:
            pending = []
            nxt = 0
            while nxt < len(batches) or pending:
                while nxt < len(batches) and len(pending) < 2:   # two batches of raw arrays in flight
                    pending.append([pool.submit(self._read, int(i)) for i in batches[nxt]])
                    nxt += 1
                samples = [self._sample(f.result()) for f in pending.pop(0)]
                xs = torch.stack([s[0] for s in samples])
                if samples[0][1] is None:
                    yield xs
                else:
                    yield xs, torch.stack([s[1] for s in samples])


class InputData:
    def __init__(self, reader=None, data_lists_train=None, data_lists_valid=None, data_lists_test=None,
                 idx_x_modalities=None, idx_y_modalities=None, x_processing=None, batch_size=1, num_workers=1,
                 transform_kwargs=None, device='cuda', shuffle_seed=None):
        self.reader = reader or (lambda x: x)
        self.data_lists_train, self.data_lists_valid, self.data_lists_test = data_lists_train, data_lists_valid, data_lists_test
        self.idx_x_modalities, self.idx_y_modalities = idx_x_modalities, idx_y_modalities
        self.x_processing = x_processing
        self.batch_size, self.num_workers = batch_size, num_workers
        self.transform_kwargs = transform_kwargs
        self.device = torch.device(device)
        self.shuffle_seed = shuffle_seed
        self.rank, self.world = 0, 1
        assert self.idx_x_modalities is not None

    def set_shard(self, rank, world):
        """One process per GPU: disjoint shards of every TRAINING and VALIDATION epoch.  The epoch permutations must be the same on
        every rank, so an unseeded shuffle (shuffle_seed=None: fresh OS entropy per process) is pinned to seed 0 here.  The test
        flow is never sharded: testing() names its outputs by the position in the test list (train_test.py:383-426), so a rank
        that runs it must see the whole list."""
        self.rank, self.world = int(rank), int(world)
        if self.world > 1 and self.shuffle_seed is None:
            self.shuffle_seed = 0

    def _get_flow(self, data_lists, shuffle=False, transform_kwargs=None, sharded=False):
        return _Flow(self, data_lists, shuffle, transform_kwargs, sharded)

    def get_train_flow(self, shuffle=True):
        return self._get_flow(self.data_lists_train, shuffle=shuffle, transform_kwargs=self.transform_kwargs, sharded=True)

    def get_valid_flow(self):
        return self._get_flow(self.data_lists_valid, sharded=True)

    def get_test_flow(self):
        return self._get_flow(self.data_lists_test)

    def _get_num_batches(self, data, sharded=False):
        world = self.world if sharded else 1
        return 0 if data is None else int(math.ceil(len(data[0]) // world / self.batch_size))

    def get_train_num_batches(self):
        return self._get_num_batches(self.data_lists_train, True)

    def get_valid_num_batches(self):
        return self._get_num_batches(self.data_lists_valid, True)

    def get_test_num_batches(self):
        return self._get_num_batches(self.data_lists_test)

    def _get_image_size(self, data):
        return None if data is None else np.asarray(self.reader(data[0][0])).shape

    def get_train_image_size(self):
        return self._get_image_size(self.data_lists_train)

    def get_valid_image_size(self):
        return self._get_image_size(self.data_lists_valid)

    def get_test_image_size(self):
        return self._get_image_size(self.data_lists_test)

    def get_num_x_modalities(self):
        return len(self.idx_x_modalities)

    def get_num_y_modalities(self):
        return 0 if self.idx_y_modalities is None else len(self.idx_y_modalities)
