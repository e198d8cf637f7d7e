"""View sharding across ranks.

Within one pass of the reference's schedule (main.cpp:452-483 / 486-508) every reference view
depends only on shared read-only data (images, cameras) and on the previous pass's maps, so views
shard embarrassingly: one process per GPU, view k -> rank k % world_size.  No collective is needed
on the data path; shared image/camera buffers are broadcast once per scene (RCCL over xGMI) before
the timed region.
"""


def views_for_rank(num_views, rank, world_size):
    """Round-robin: identical to a 1-rank run restricted to these indices (views are independent
    within a pass), and balanced to within one view."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    return list(range(rank, num_views, world_size))


def owner_of(view, world_size):
    return view % world_size
