// gymrs_aql.h -- the engine's own AQL dispatcher: chains of per-step launches written straight into an HSA queue.
//
// Why (measured on MI355X, tools/aql/aql_probe.cpp, profiles/r03_aql_probe.log): the HIP runtime gives EVERY launch an
// agent-scope acquire and an agent-scope RELEASE fence (AQL header 0xb02).  The release is an L2 write-back at the end of
// the kernel, and for a kernel of the step's shape (17 B read + 21 B written per lane, 2^20 lanes) it costs 1.6-1.9 us per
// launch: 7.0 us per launch with both fences, 5.1 with the acquire alone -- the acquire (L1 / scalar-cache invalidate) is
// free.  A chain of gymrs_step_many launches does not need the write-back between its links: tile i is stepped by workgroup
// i in every launch, workgroups are dealt round-robin to the XCDs, so the lines launch t leaves dirty in an XCD's L2 are
// read by launch t + 1 on that very XCD.  Only whoever reads the arrays AFTER the chain (another kernel, a copy, the host)
// needs them written back, so the chain's last packet carries a system-scope release.  HIP has no way to say that; AQL has:
// the fence scopes are two fields of the packet header.  As a by-product a launch costs the host ~0.3 us instead of 2.5-3.5.
//
// The dispatcher is an ALTERNATIVE SUBMISSION PATH for the same kernels (the same templates compiled once more into a
// stand-alone code object, gymrs_step_aql.hip), not another implementation: everything it cannot do -- or any device on which
// its self-check fails -- goes through HIP launches as before.  GYMRS_AQL=0 in the environment switches it off.
//
// What the chain's launches must NOT do is mark their accesses non-temporal: a streamed line leaves the L2 early, and the
// chain lives off the lines the previous launch left there (measured: 5.40 us per 2^20-lane CartPole step with plain accesses,
// 6.2 rising to 7.5 with the hint; HIP launches: 6.41 with the hint, 7.13 without).  The engine picks the hints per path.
//
// Ordering against the engine's HIP stream (gymrs_step_many is asynchronous ON THAT STREAM):
//   begin: a one-wave kernel on the stream stores seq into in_flag, behind whatever is enqueued there; the chain's first packet
//          is a one-wave kernel that waits for in_flag >= seq (bounded: ~10 s, then it reports and lets the chain run);
//   end:   after a packet whose end-of-kernel system-scope release writes back what the chain left dirty, the chain's last
//          packet stores seq into out_flag; a one-wave kernel on the stream waits for that, so everything enqueued on the
//          stream later -- and gymrs_sync -- comes after the chain.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>

namespace gymrs {

struct AqlKernel {
    uint64_t object = 0; // kernel descriptor address
    uint32_t kernarg_bytes = 0, group_bytes = 0, private_bytes = 0;
};

class AqlChain;

// One chain object per engine (its own HSA queue).  nullptr + *why when the path is not available: no large-BAR access to
// device memory, no stream memory operations, a failed self-check, GYMRS_AQL=0, ...
AqlChain* aql_create(int hip_device, std::string* why);
// parks the object (queue, rings, flags) for the next aql_create on the device: hardware queues are created once, not per engine
// discard: an error path (a failed hand-over, a tripped XCD check, a stream that could not be waited for) -- the queue is given back, not parked.
void aql_destroy(AqlChain* c, bool discard = false);
bool aql_kernel(AqlChain* c, const char* name, AqlKernel* out);
// the embedded stand-alone code object itself (developer experiment: the same binary launched through the HIP runtime's queue, gymrs_engine.hip)
const void* aql_code_blob(size_t* bytes);

// Largest kernel-argument block one dispatch may carry.
constexpr size_t kAqlKernargSlot = 512;

// begin -> dispatch ... dispatch -> end.  `stream` is the engine's HIP stream.  On failure *err says why and the chain object
// stays usable only for aql_destroy (the engine then falls back to HIP launches for good).
bool aql_begin(AqlChain* c, hipStream_t stream, std::string* err);
// grid_workitems = workgroups * workgroup_size.  Kernel arguments are copied.
// release: the packet also carries an agent-scope RELEASE (HIP's own header) instead of the chain's acquire-only one.
// system_acquire: the first launch behind aql_begin -- the acquire that FOLLOWS the hand-over (gymrs_aql.hip says why the opening packet's own is not enough).
bool aql_dispatch(AqlChain* c, const AqlKernel& k, uint32_t grid_workitems, uint32_t workgroup_size, const void* kernarg, size_t bytes,
                  std::string* err, bool release = false, bool system_acquire = false);
bool aql_end(AqlChain* c, hipStream_t stream, std::string* err);
// != 0 once a chain's first packet gave up waiting for the stream (checked by gymrs_sync); cleared by the call
uint32_t aql_take_error(AqlChain* c);
// the chains of this object use the synchronous hand-over (the host waits on both sides): see aql_create
bool aql_is_synchronous(const AqlChain* c);
// Decides, for this stream, between the asynchronous hand-over and the synchronous one (AqlChain::calibrated_for).  Cheap after
// the first call per stream (it looks again only when another chain object has been CREATED on the device since).  Only a stream
// the engine owns is timed (~3 ms of probe chains, hipStreamSynchronize on that stream); a caller-provided stream is never waited
// for: its chains use the asynchronous hand-over.  Returns a description when it decided anew (valid until the next call), NULL otherwise.
const char* aql_calibrate(AqlChain* c, hipStream_t stream, bool own_stream);
// [8] device words of this chain object for StepArgs::xcc_table: the first step launch of a chain records where its workgroups run,
// the later launches of the chain compare (gymrs_kernels.h says why the table is per chain and not per queue or per device).
uint32_t* aql_xcc_table(const AqlChain* c);
// the number of the chain that is open (aql_begin counts; 24 bits): the tag of its table entries.  It belongs to the chain OBJECT, which
// outlives engines (aql_destroy parks it for the next engine of the device), so no two chains that ever wrote one table share a tag.
uint32_t aql_chain_number(const AqlChain* c);

} // namespace gymrs
