"""Batch loader on THREADS for datasets whose per-batch work is a ``collate_fn`` of GIL-free array copies
(lirec_amd.features.PiecesDataset): the ``torch.utils.data.DataLoader`` protocol ``training()`` / ``testing()`` use
(mlp/train.py:33-37, mlp/test.py:18-22 -- dataset, batch_size, shuffle / sampler, num_workers, drop_last; ``len()``; one
pass per ``iter()``), with ``num_workers`` threads instead of worker processes.

Why not worker processes here: they are forked from a process that has initialised the GPU, and with eight of them alive
the same train step took 36 ms of GPU time instead of 1 ms on the MI355X box (profiles/r03_training_entry.txt); a batch of
piece tables is also 2.4 MB that would cross a pipe and be pinned again on the other side.  Threads hand over the pinned
tensors the collate wrote.  Batches come out in sampler order whatever the thread count."""
from __future__ import annotations

import queue
import threading

import torch


class ThreadedLoader:
    def __init__(self, dataset, batch_size=1, shuffle=False, sampler=None, num_workers=0, collate_fn=None, drop_last=False,
                 prefetch=4):
        self.dataset, self.collate_fn = dataset, collate_fn or torch.utils.data.default_collate
        if sampler is None:
            sampler = torch.utils.data.RandomSampler(dataset) if shuffle else torch.utils.data.SequentialSampler(dataset)
        self.batch_sampler = torch.utils.data.BatchSampler(sampler, batch_size, drop_last)
        self.num_workers, self.prefetch = max(int(num_workers), 0), max(int(prefetch), 1)

    def __len__(self):
        return len(self.batch_sampler)

    def _make(self, idx):
        return self.collate_fn([self.dataset[i] for i in idx])

    def __iter__(self):
        if self.num_workers == 0:
            for idx in self.batch_sampler:
                yield self._make(idx)
            return
        todo = list(self.batch_sampler)                     # (the sampler is drawn on the caller's thread: its RNG state is the caller's)
        n, depth = len(todo), self.num_workers * self.prefetch
        slots = [queue.Queue(maxsize=1) for _ in range(n)]  # one slot per batch: ordered hand-over
        nxt, lock, stop = [0], threading.Lock(), threading.Event()
        room = threading.Semaphore(depth)                   # at most `depth` batches built ahead of the consumer

        def work():
            while not stop.is_set():
                room.acquire()
                with lock:
                    k = nxt[0]
                    nxt[0] += 1
                if k >= n or stop.is_set():
                    return
                try:
                    slots[k].put(('ok', self._make(todo[k])))
                except BaseException as e:                  # handed to the consumer, which re-raises it in order
                    slots[k].put(('err', e))
        threads = [threading.Thread(target=work, daemon=True) for _ in range(self.num_workers)]
        for t in threads:
            t.start()
        try:
            for k in range(n):
                kind, val = slots[k].get()
                slots[k] = None
                room.release()
                if kind == 'err':
                    raise val
                yield val
        finally:
            stop.set()
            for _ in threads:
                room.release()
