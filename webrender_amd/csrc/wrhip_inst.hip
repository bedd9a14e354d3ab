// One group of raster kernel instantiations (wrhip_inst.h); compiled once per group with -DWR_INST_GROUP=n.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <map>
#include <algorithm>
#include <time.h>
#include <chrono>
#include <thread>
#include <pthread.h>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <functional>
#include <tuple>
#define WR_INST_ONLY 1
#include "wrhip_types.h"
#include "wrhip_rt.h"
#include "wrhip_kernels.h"
#include "wrhip_inst.h"
#define WR_INST_CAT_(a, b) a##b
#define WR_INST_CAT(a, b) WR_INST_CAT_(a, b)
#define WR_INST_DEFINE(K, SIG, ...) template __global__ void K<__VA_ARGS__> SIG;
WR_INST_CAT(WR_INST_, WR_INST_GROUP)(WR_INST_DEFINE)
