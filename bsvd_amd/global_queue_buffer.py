"""Segment-to-segment state for MIMO (segmented whole-clip) inference of the TSN network: a FIFO of per-layer "past"
slices plus the current segment index and look-ahead length.  Same role and function names as
/root/reference/Experimental_root/models/global_queue_buffer.py:9-46 so that ``denoise_seq`` drives it identically;
the values queued here are compact device slices [H,W,fold] produced by ``bsvd_halo_pack``.
Module-global like the reference's (not re-entrant; one clip at a time)."""
from collections import deque

_queue = deque()
_future_buf_len = 0
_batch_index = -1


def _init(future_buffer_len):
    global _future_buf_len, _batch_index
    _queue.clear()
    _future_buf_len = int(future_buffer_len)
    _batch_index = -1


def _clean():
    _queue.clear()


def put(value):
    _queue.append(value)


def get():
    return _queue.popleft()


def qsize():
    return len(_queue)


def get_future_buffer_length():
    return _future_buf_len


def set_future_buffer_length(future_buffer_len):
    global _future_buf_len
    _future_buf_len = int(future_buffer_len)
    return _future_buf_len


def get_batch_index():
    return _batch_index


def set_batch_index(idx):
    global _batch_index
    _batch_index = int(idx)
