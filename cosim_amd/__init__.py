"""cosim_amd: MI355X-native batched rollout engine behind the cosim env API."""
import os as _os

# Range launches (BatchedEnv(ranges=S) / cosim_set_param "ranges") put S streams of the engine beside the caller's: the HIP runtime
# maps streams onto 4 hardware queues by default and streams that share a queue run their kernels one after the other (measured:
# four range streams then step at 7.7 M env-steps/s instead of 13.9 M).  Read by the runtime at its first HIP call, so set here,
# before anything of this package can touch the GPU; an explicit value in the environment wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
