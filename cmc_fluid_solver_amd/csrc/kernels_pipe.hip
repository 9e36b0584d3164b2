// placeholder: the wave-pipelined sweep kernel is added after the baseline is parity-green
#include "fs3d_common.h"
template <typename R> bool launch_sweep_pipe(fs3d_ctx *, int, const SweepParams<R> &) { return false; }
template bool launch_sweep_pipe<float>(fs3d_ctx *, int, const SweepParams<float> &);
template bool launch_sweep_pipe<double>(fs3d_ctx *, int, const SweepParams<double> &);
