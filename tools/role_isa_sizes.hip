// tools/role_isa_sizes.hip -- measurement aid, not part of the library: every role of the wave pipeline as a kernel of its own,
// for static ISA sizes (hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only tools/role_isa_sizes.hip;
// profiles/archive/r02_pipeline_role_isa_static.md)
#include <hip/hip_runtime.h>
#include "../icer_compression_amd/csrc/assemble_core.hpp"
#include "../icer_compression_amd/csrc/coder_core.hpp"
using namespace icer;
#define ROLE(NAME, BODY) extern "C" __global__ void __launch_bounds__(64) NAME(UnitArgs a, uint32_t n, uint32_t *out) { __shared__ CoderShared s; BODY }
ROLE(role_pixel, { PixelWave pw; pixel_wave_run(s, a, pw, 0, n, 0u, 1u); })
ROLE(role_count, { CountWave cs; count_wave_run(s, a, cs, 0, n, 1u); })
ROLE(role_compact, { compact_wave_run(s, a, 0, n); })
ROLE(role_walker, { WalkWave ww; walk_wave_init(s, ww); out[0] = walk_wave_run(s, a, ww, n, ~0u); })
ROLE(role_golomb_fused, { GolombWave gw; golomb_wave_init(gw); out[0] = golomb_wave_run(s, a, gw, n, ~0u, 0u, 0u); })
ROLE(role_golomb_state, { GolombWave gw; golomb_wave_init(gw); out[0] = golomb_state_run(s, a, gw, n, ~0u); })
ROLE(role_golomb_worker, { GolombWave gw; golomb_wave_init(gw); out[0] = golomb_wave_run(s, a, gw, n, ~0u, 1u, 2u); })
ROLE(role_records, { RecordsWave rw; records_wave_run(s, a, rw, ~0u); })
ROLE(role_drain, { drain_wave_run(s, a, ~0u); })
ROLE(role_merge, { out[0] = merge_wave_run(s, a, 0, n) ? merge_wave_finish(s, a) : 0u; })
