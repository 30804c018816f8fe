"""Image-tile partition across the GPUs of one node and the gather of the
tiles to rank 0 (north_star subsystem 4).  The scene is replicated per GPU;
each rank renders the tiles it owns (mipt_render_params.part_index/part_count)
into its own frame buffer; ONE collective per step moves the owned pixels to
rank 0: torch.distributed.gather, which is RCCL over xGMI with backend "nccl".
No all-reduce: a ring all-reduce of full frames would be per-link bound and
move N times the bytes for nothing.

Ownership rule (same as the C library, api_device.hip): tile (tx, ty) of
tile_w x tile_h pixels belongs to part (tx + 3*ty) % parts.
"""
import numpy as np

TILE_W, TILE_H = 64, 8


def tile_owner_map(width, height, parts, tile_w=TILE_W, tile_h=TILE_H):
    """(height, width) int array: the part that renders each pixel"""
    tiles_x, tiles_y = (width + tile_w - 1) // tile_w, (height + tile_h - 1) // tile_h
    ty, tx = np.meshgrid(np.arange(tiles_y), np.arange(tiles_x), indexing="ij")
    owner_of_tile = (tx + 3 * ty) % max(1, parts)
    py, px = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    return owner_of_tile[py // tile_h, px // tile_w]


def owned_pixel_indices(width, height, parts, tile_w=TILE_W, tile_h=TILE_H):
    """list over parts of sorted flat pixel indices (row-major, origin bottom-left like the frame buffer)"""
    owner = tile_owner_map(width, height, parts, tile_w, tile_h).reshape(-1)
    return [np.flatnonzero(owner == r) for r in range(max(1, parts))]


class TileGather:
    """Pre-allocated buffers + index tensors for the per-step gather."""

    def __init__(self, width, height, rank, world, device, tile_w=TILE_W, tile_h=TILE_H, always=False):
        """always: run the collective even in a world of one rank (rehearsing the RCCL path on a one-GPU box)"""
        import torch
        self.rank, self.world, self.always = rank, world, always
        self.idx = [torch.from_numpy(i).to(device) for i in owned_pixel_indices(width, height, world, tile_w, tile_h)]
        self.max_own = max(int(i.numel()) for i in self.idx)
        self.send = torch.zeros(self.max_own * 3, dtype=torch.float32, device=device)
        self.recv = ([torch.zeros(self.max_own * 3, dtype=torch.float32, device=device) for _ in range(world)]
                     if rank == 0 else None)

    def bytes_per_step(self):
        """bytes rank 0 receives per gather"""
        return (self.world - 1) * self.max_own * 12

    def gather(self, rgb_flat):
        """rgb_flat: this rank's frame buffer as a flat float tensor of width*height*3; on rank 0 the pixels
        of the other ranks are written into it"""
        import torch.distributed as dist
        if self.world == 1 and not self.always:
            return
        rgb3 = rgb_flat.view(-1, 3)
        n = self.idx[self.rank].numel()
        self.send[: n * 3] = rgb3.index_select(0, self.idx[self.rank]).reshape(-1)
        dist.gather(self.send, self.recv, dst=0)
        if self.rank == 0:
            for r in range(1, self.world):
                m = self.idx[r].numel()
                rgb3.index_copy_(0, self.idx[r], self.recv[r][: m * 3].view(-1, 3))

    def gather_async(self, rgb_flat):
        """gather() on a stream of its own, beside whatever the caller enqueues next on the current stream (the next
        frame's kernels): the pack waits for the frame by an event, the collective and rank 0's unpack follow on the side
        stream, and the current stream waits only for the pack (which reads rgb_flat) -- what mipt_node_render does with
        its transfer stream (api_node.hip).  Rank 0's unpack writes the OTHER ranks' pixels of rgb_flat while the next
        frame's kernels write rank 0's own: disjoint elements.  finish() (or a device-wide synchronize) before the frame is
        read.  CPU tensors (gloo tests) take the synchronous path."""
        import torch
        import torch.distributed as dist
        if self.world == 1 and not self.always:
            return
        if not rgb_flat.is_cuda:
            return self.gather(rgb_flat)
        dev = rgb_flat.device
        cur = torch.cuda.current_stream(dev)
        if getattr(self, "side", None) is None:
            self.side = torch.cuda.Stream(device=dev)
            self.ev_rendered = torch.cuda.Event()
            self.ev_packed = torch.cuda.Event()
        self.ev_rendered.record(cur)
        rgb3 = rgb_flat.view(-1, 3)
        n = self.idx[self.rank].numel()
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ev_rendered)
            # send / recv are touched on the side stream only, so successive steps are ordered there
            self.send[: n * 3] = rgb3.index_select(0, self.idx[self.rank]).reshape(-1)
            self.ev_packed.record(self.side)
            work = dist.gather(self.send, self.recv, dst=0, async_op=True)
            work.wait()  # the side stream (not the host) waits for the collective
            if self.rank == 0:
                for r in range(1, self.world):
                    m = self.idx[r].numel()
                    rgb3.index_copy_(0, self.idx[r], self.recv[r][: m * 3].view(-1, 3))
        cur.wait_event(self.ev_packed)  # the next frame may overwrite this rank's pixels once the pack has read them

    def finish(self):
        """the current stream waits for everything gather_async has enqueued"""
        import torch
        if getattr(self, "side", None) is not None:
            torch.cuda.current_stream(self.side.device).wait_stream(self.side)

    def gather_rgb8(self, rgb8_flat):
        """the same collective for a frame that has already been converted to 8 bits per channel on the device
        (Context.pack_rgb8_device: Image::save's conversion): rgb8_flat is this rank's frame as a flat uint8 tensor
        of width*height*3; rank 0 receives a quarter of the bytes of gather()"""
        import torch
        import torch.distributed as dist
        if self.world == 1 and not self.always:
            return
        if not hasattr(self, "send8"):
            dev = self.send.device
            self.send8 = torch.zeros(self.max_own * 3, dtype=torch.uint8, device=dev)
            self.recv8 = ([torch.zeros(self.max_own * 3, dtype=torch.uint8, device=dev) for _ in range(self.world)]
                          if self.rank == 0 else None)
        rgb3 = rgb8_flat.view(-1, 3)
        n = self.idx[self.rank].numel()
        self.send8[: n * 3] = rgb3.index_select(0, self.idx[self.rank]).reshape(-1)
        dist.gather(self.send8, self.recv8, dst=0)
        if self.rank == 0:
            for r in range(1, self.world):
                m = self.idx[r].numel()
                rgb3.index_copy_(0, self.idx[r], self.recv8[r][: m * 3].view(-1, 3))
