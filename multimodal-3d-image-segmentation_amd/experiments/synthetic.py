"""Synthetic stand-in for the reference's InputData (experiments/data_io/input_data.py:15-151):
the methods the training loop uses, fed by seeded random volumes instead of NIfTI files."""
import math

import torch


class SyntheticInputData:
    def __init__(self, image_size, in_channels, num_labels, batch_size=1, num_train=4, num_valid=2, seed=1234,
                 generator=None, num_test=2):
        self.image_size = tuple(image_size)
        self.in_channels, self.num_labels = in_channels, num_labels
        self.batch_size, self.num_train, self.num_valid = batch_size, num_train, num_valid
        self.seed = seed
        self.num_test = num_test
        self.data_lists_test = [[f'synthetic_{i}'] for i in range(num_test)]
        self._make = generator or self._default_sample
        self.rank, self.world = 0, 1

    def set_shard(self, rank, world):
        """Data-parallel training: this process iterates samples rank, rank + world, ... of every epoch's order (the
        permutation is drawn from the shared seed, so the shards are disjoint); the tail that does not divide evenly is
        dropped so that every rank runs the same number of batches (a collective per batch would otherwise hang).  Training and
        validation flows only: the test flow always yields every sample (testing() names outputs by list position)."""
        self.rank, self.world = int(rank), int(world)

    def _default_sample(self, index):
        g = torch.Generator().manual_seed(self.seed + index)
        x = torch.randn((self.in_channels,) + self.image_size, generator=g)
        y = torch.randint(0, self.num_labels, (1,) + self.image_size, generator=g).float()
        return x, y

    def _flow(self, first, count, shuffle, epoch_seed=0, sharded=True):
        order = list(range(first, first + count))
        if shuffle:
            g = torch.Generator().manual_seed(self.seed + 7919 + epoch_seed)
            order = [order[i] for i in torch.randperm(count, generator=g).tolist()]
        if sharded and self.world > 1:
            order = order[:(count // self.world) * self.world][self.rank::self.world]
            count = len(order)
        for i in range(0, count, self.batch_size):
            xs, ys = zip(*[self._make(j) for j in order[i:i + self.batch_size]])
            yield torch.stack(xs), torch.stack(ys)

    # -- the part of the InputData interface used by training() / testing() ------------------------------
    def get_train_flow(self, shuffle=True):
        return _Reiterable(lambda: self._flow(0, self.num_train, shuffle))

    def get_valid_flow(self):
        return _Reiterable(lambda: self._flow(self.num_train, self.num_valid, False))

    def get_test_flow(self):
        return _Reiterable(lambda: self._flow(self.num_train + self.num_valid, self.num_test, False, sharded=False))

    def get_test_num_batches(self):
        return int(math.ceil(self.num_test / self.batch_size))

    def get_train_num_batches(self):
        return int(math.ceil(self.num_train // self.world / self.batch_size))

    def get_valid_num_batches(self):
        return int(math.ceil(self.num_valid // self.world / self.batch_size))

    def get_train_image_size(self):
        return self.image_size


class _Reiterable:
    """A DataLoader-like object: iterating it again restarts the flow (train_test.py:146,189)."""

    def __init__(self, factory):
        self._factory = factory

    def __iter__(self):
        return self._factory()
