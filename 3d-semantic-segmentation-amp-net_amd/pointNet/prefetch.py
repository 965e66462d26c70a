"""Pinned-memory prefetch of the input pipeline: the collated batch of step i + 1 travels to the GPU on a copy stream
while step i computes, so a train_loop call no longer pays the 51 MB upload (42 MB of points + 9 MB of labels at B = 64) on its
critical path.  The reference moves every window to the device inside the step, nine synchronous copies
(train_pointnet-attention.py:396-408).

    for data in DevicePrefetcher(train_dataloader, device):      # DataLoader(..., pin_memory=True, collate_fn=collate_seq_padd)
        train_loop(data, ...)                                      # sees device tensors: its own upload is a no-op

Only tensors move; lists (file names) pass through.  The batch handed out is safe to use on the current stream (event wait +
record_stream), and the loader's pinned host buffers are released as soon as their copy has been issued."""
import torch


_COPY_STREAMS = {}


def copy_stream(device):
    """ONE copy stream per device for the life of the process.  torch's caching allocator keeps a pool per stream: with a new stream per
    epoch the ~30 upload blocks of an epoch (20 MB each) could never be reused by the next one -- 0.6 GB of new hipMalloc calls per epoch,
    seconds of stall each time and memory that only grows (bench.py train_att_epoch: device_allocator)."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _COPY_STREAMS:
        _COPY_STREAMS[key] = torch.cuda.Stream(device)
    return _COPY_STREAMS[key]


class DevicePrefetcher:
    def __init__(self, loader, device, depth=1):
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, int(depth))
        self.stream = copy_stream(self.device)

    def __len__(self):
        return len(self.loader)

    def _upload(self, batch):
        with torch.cuda.stream(self.stream):
            moved = tuple(x.to(self.device, non_blocking=True) if (torch.is_tensor(x) or hasattr(x, "record_stream")) else x for x in batch)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return moved, ev

    def __iter__(self):
        queue = []
        it = iter(self.loader)
        cur = torch.cuda.current_stream(self.device)
        for batch in it:
            queue.append(self._upload(batch))
            if len(queue) > self.depth:
                yield self._hand_out(queue.pop(0), cur)
        while queue:
            yield self._hand_out(queue.pop(0), cur)

    @staticmethod
    def _hand_out(item, cur):
        moved, ev = item
        cur.wait_event(ev)
        for x in moved:
            if (torch.is_tensor(x) or hasattr(x, "record_stream")) and x.is_cuda:      # tensors and collate_fns.RaggedBatch
                x.record_stream(cur)                 # allocated on the copy stream, consumed on the compute stream
        return moved
