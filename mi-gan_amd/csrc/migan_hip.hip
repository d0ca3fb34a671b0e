// libmigan_hip.so translation unit: gfx950 kernels + plan + C ABI (include/migan_hip.h, include/comodgan_hip.h).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared migan_hip.hip -o libmigan_hip.so
#include "migan_rt_hip.h"
#include "migan_kernels.hpp"
#include "migan_table.hpp"
#include "migan_pipe.hpp"
#include "migan_wide2.hpp"
#include "comodgan_kernels.hpp"
#include "migan_host.hpp"
#include "comodgan_host.hpp"
