// Host object model behind the opaque embree3 handles: Device, Buffer, Geometry, Scene.
// Mirrors the roles of kernels/common/{device,state,buffer,geometry,scene_triangle_mesh,scene_subdiv_mesh,scene}.*
// of the reference, reduced to what the traversal hot path and its builders need.
#pragma once
#include <map>
#include <memory>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_map>

#include "accel.h"
#include "rt_common.h"

namespace rtamd {

struct Scene;

// ---------------------------------------------------------------------------------------------------
// Device: config string, error state, HIP device + stream, staging buffers.
// ---------------------------------------------------------------------------------------------------
struct Device : RefCounted
{
  // config (reference: State::parse, kernels/common/state.cpp:241-430)
  std::string tri_accel = "default";
  std::string subdiv_accel = "default";
  int verbose = 0;
  int gpu = 0;            // HIP device ordinal ("gpu=" key; falls back to env RTAMD_GPU, LOCAL_RANK is NOT read here)
  int numThreads = 0;     // accepted, used only for host-side builders
  int benchmark = 0;
  int keepGrids = 0;      // "keep_grids=1": keep the tessellated vertex grids for inspection (tests)

  // error state (reference: Device::process_error device.cpp:258-286, getDeviceErrorCode :250-256)
  std::mutex errMutex;
  std::unordered_map<uint64_t, RTCError> threadErrors; // per calling thread, first error wins
  RTCErrorFunction errorFn = nullptr;
  void* errorFnUser = nullptr;
  RTCMemoryMonitorFunction memFn = nullptr;
  void* memFnUser = nullptr;

  // HIP state.  One GpuShard per GPU the device traces on ("gpu=<ordinal>" -> one shard; "gpus=0-7" / "gpus=0:2:5" ->
  // several: the accel is replicated on every shard at commit and one host-pointer rtcIntersect1M / rtcOccluded1M call is
  // split into contiguous ray ranges, one per shard, each with its own stream and staging buffers and a disjoint D2H into
  // the caller's records - SURVEY.md section 8e; no collective).  The same ordinal may be listed twice ("gpus=0:0"): two
  // logical shards on one GPU, which is how the sharded path is tested on a one-GPU box.
  struct LaunchCtx
  {
    void* queues = nullptr;  // TRACE_QUEUES heads, one 128-byte line each
    void* spill = nullptr;
    size_t spillBytes = 0;
    void* survivors = nullptr; // root cull pre-pass: per-queue survivor lists (4 bytes per ray of the largest batch so far)
    size_t survivorsBytes = 0;
    hipEvent_t done = nullptr;
    bool used = false;
    hipStream_t stream = nullptr; // stream of the launch that last used this context
  };
  static const int NUM_LAUNCH_CTX = 8;
  struct GpuShard
  {
    int ordinal = 0;
    int numCUs = 256;
    hipStream_t stream = nullptr;
    bool ownsStream = false;
    // staging (host-pointer path): pinned host mirror + device batch, grown on demand
    void* stageHost = nullptr;
    void* stageDev = nullptr;
    size_t stageBytes = 0;
    // pipelined host-pointer batches (rt_trace.cpp trace_host_pipelined): chunks alternate between two internal streams, so that
    // the upload of one chunk runs under the traversal and the download of the previous one (PCIe is full duplex); one event per chunk
    hipStream_t pipeStream[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> pipeEvents;
    // Per-launch scratch (work-queue heads + LDS-stack overflow area).  A ring of contexts, so that batches enqueued on
    // DIFFERENT streams (rtcamdSetDeviceStream between calls) can be in flight together: the drain of one batch - a few
    // deep rays keeping waves alive - then overlaps the start of the next.  A context is reused only after the kernel
    // that last used it has finished (stream-side wait on its event, no host block).
    LaunchCtx launchCtx[NUM_LAUNCH_CTX];
    unsigned nextCtx = 0;
    // serialises {pick a launch context, zero its queue heads, launch, record its event}: rtcIntersect1M / rtcOccluded1M on
    // device-resident batches are callable from several threads at once, like the reference's (rtcore.cpp:403-432)
    std::mutex seqMutex;
    void* countersDev = nullptr; // wave log of the instrumented twin (one counted batch at a time, under launchMutex)
    // One word of host-mapped pinned memory the kernels set when a traversal-stack entry had to be dropped (the overflow
    // area is sized from the tree depth, so this cannot happen for a tree the builder made; if it ever does, the call
    // raises RTC_ERROR_UNKNOWN instead of silently missing a subtree).
    uint32_t* overflowHost = nullptr;
    uint32_t* overflowDev = nullptr;
    void use() const;
    void ensureStaging(size_t bytes);
    // picks the next context, makes `stream` wait for its previous user; *busyOther = OTHER streams with unfinished launches.
    // Call with seqMutex held.
    LaunchCtx& acquireLaunchCtx(size_t spillBytesNeeded, unsigned* busyOther = nullptr, hipStream_t onStream = nullptr);
    void checkOverflow(); // throws RTC_ERROR_UNKNOWN (and clears the flag) when a kernel reported a dropped stack entry
  };
  std::vector<std::unique_ptr<GpuShard>> shards; // empty for gpu=none
  std::vector<int> gpuList;                      // "gpus=" ordinals in the order given
  GpuShard& primary() { return *shards[0]; }
  std::mutex launchMutex; // serialises host-pointer batches (they share the staging buffers) and counted batches

  // Host threads for the staging copies of large host-pointer batches (gather of the caller's records into pinned memory, scatter of
  // tfar + hit back): one thread moves ~10 GB/s, PCIe 5 x16 ~55 GB/s each way.  Started with the first such batch; the caller works too.
  struct HostPool
  {
    // While a pipelined batch is running (`hot`) the helpers spin on `ticket` instead of sleeping: a part of a chunk is ~100 us of
    // copying, a condition-variable wake-up costs 20-50 us.  ticket = (job generation << 32) | next part: a helper that is late for
    // a job sees another generation and takes nothing.
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv;
    std::atomic<bool> stop{false}, hot{false};
    std::atomic<uint64_t> ticket{0};
    std::atomic<size_t> done{0};
    std::atomic<const std::function<void(size_t)>*> job{nullptr}; // (a helper that is late may read these while the next job is being
    std::atomic<size_t> nParts{0};                                 //  set up: its generation check then fails and it takes nothing)
    void start(unsigned n);
    void begin();                                                  // helpers start polling
    void end();                                                    // helpers go back to sleep
    void run(size_t parts, const std::function<void(size_t)>& f); // f(0..parts-1), returns when all are done; the caller works too
    bool take(uint64_t gen, size_t n, size_t& i);
    ~HostPool();
  };
  HostPool hostPool;
  uint32_t tuneZeroCopyMax = 512;      // env RTAMD_ZEROCOPY_MAX: host batches up to this many rays are traced in place in pinned host memory (0 = never)
  uint32_t tuneHostThreads = 0;        // env RTAMD_HOST_THREADS (0: min(8, hardware threads / 2))
  uint32_t tunePipeMinRays = 16384;    // env RTAMD_PIPE_MIN: host-pointer batches from this size on are pipelined in chunks
  uint32_t tunePipeChunk = 0;          // env RTAMD_PIPE_CHUNK: rays per chunk; 0 = by batch size (rt_trace.cpp, measured optima)

  // Call combiner for small host-pointer calls (rtcIntersect1 / rtcOccluded1 / short 1M streams from many threads):
  // whoever finds the device idle becomes the leader and traces everything that is pending - its own call and the
  // calls that queued up behind the previous launch - as ONE batch; the others sleep until their record is done.
  struct SmallCall
  {
    struct Scene* scene;
    char* base;
    uint32_t M;
    size_t stride;
    bool occluded;
    uint32_t instID;
    bool done = false;
    RTCError error = RTC_ERROR_NONE;
    std::string message;
  };
  // Persistent consumer for calls of up to 64 rays (trace_service.hip.h, rt_trace.cpp service_trace): env RTAMD_SERVICE / config key service=1
  struct Service;
  Service* service = nullptr;
  std::mutex serviceMutex; // creation / restart of the service kernel
  uint32_t tuneService = 0;
  std::mutex combMutex;
  std::condition_variable combCv;
  std::vector<SmallCall*> combPending;
  bool combBusy = false;
  std::atomic<bool> combHold{false}; // test hook: rtcamdDebugHoldCombiner
  std::atomic<uint64_t> statLaunches{0};      // traversal kernel launches
  std::atomic<uint64_t> statServiceCalls{0};  // calls answered by the persistent service kernel
  std::atomic<uint64_t> statCombinedCalls{0}; // calls that went through the combiner
  std::atomic<uint64_t> statCombinedBatches{0}; // batches the combiner formed out of them
  uint32_t tuneRefillBatch = 8; // env RTAMD_REFILL_BATCH
  uint32_t tuneOctLeaf = 0xFFFFFFFFu; // auto: 16 for triangle leaves, 24 for grid cells (measured optima);    // env RTAMD_OCT_LEAF (trace_loop.hip.h, octet leaf step; leaves that have one)
  uint32_t tuneOctSteps = 2;    // env RTAMD_OCT_STEPS
  uint32_t tuneCbvhForm = 2;    // env RTAMD_CBVH_FORM: 0 quad form, 1 one ray per lane, 2 by the context's coherent flag (rt_trace.cpp)
  // env RTAMD_CULL=1: root cull pre-pass in front of the lane kernel (trace_cull.hip.h).  OFF by default: measured on MI355X
  // (profiles/r02_cull_ab.txt) it removes 17 % of the wave instructions of a 1 M-ray batch (40.8 M -> 6.0 M + 27.8 M) but the
  // traversal kernel, fed with survivors only, takes longer (158 -> 24 + 181 us alone) and four batches in flight gain nothing
  // (cbvh.leaf +0.7 %, triangles +5 %, eager -12 %): the kernels are bound by dependent latency at three waves per SIMD, not by
  // instruction issue.
  uint32_t tuneCull = 0;
  uint32_t tuneCullMinRays = 65536; // batches below this size go straight to the traversal kernel (the extra launch costs ~5 us)
  uint32_t tuneAloneBlocksOct = 4; // env RTAMD_ALONE_BLOCKS: workgroups per CU of a batch alone on the chip, octet-only leaf kernels
  uint32_t tuneBusyBlocksOct = 2;  // env RTAMD_BUSY_BLOCKS: the same with two or more batches running on other streams
  uint32_t tuneWalkBatch = 16;  // env RTAMD_WALK_BATCH (trace_loop.hip.h, two-stage blob visits: rays parked after the frustum test before a walk pass runs;
                                // 16 = one full pass of quads; measured 8..32, profiles/r03_two_stage_ab.txt)
  uint32_t tuneOctMax = 16;     // env RTAMD_OCT_MAX (trace_loop.hip.h, octet node step), clamped to the build's TRACE_OCT_MAX in the kernel
  // kernel tuning knobs (env RTAMD_CHUNK / RTAMD_LEAF_BATCH / RTAMD_BLOCKS_PER_CU).  Measured on MI355X, 1 M-ray batches:
  // alone on the chip every setting within chunk 128-256, leaf batch 24-40, 2-3 workgroups per CU is within +-4 %; with
  // four batches in flight 256 / 32 / 2 is +15-20 % over 128 / 24 / 3 (each kernel leaves the third wave slot of a SIMD
  // to the other batches, and takes its rays in fewer, larger grabs).
  uint32_t tuneChunk = 256, tuneLeafBatch = 32, tuneBlocksPerCU = 2;
  uint32_t tuneChunkBusy = 512; // env RTAMD_CHUNK_BUSY: upper bound of the chunk of a large batch launched while two or more others run (rt_trace.cpp launch_on)
  bool tuneChunkFixed = false; // RTAMD_CHUNK given: every batch uses exactly that chunk (otherwise small batches are cut finer, rt_trace.cpp ray_chunk_for)
  // traversal skeleton: 0 lane-per-ray (trace_loop.hip.h), 1 ray pool (trace_pool.hip.h), 2 by batch size: the pool kernel's
  // steady state is 14 % faster, its drain slower - it wins from ~2.5 M rays per launch on (env RTAMD_KERNEL=lane|pool|auto)
  uint32_t tunePoolKernel = 2;
  uint32_t tunePoolMinRays = 2500000;

  explicit Device(const char* cfg);
  ~Device() override;

  void parse(const std::string& cfg);
  void setError(RTCError code, const char* msg);
  RTCError takeError();
  void useDevice() const; // hipSetDevice(first shard) for the calling thread; throws on a gpu=none device
  void synchronize();     // all shards' streams; raises a pending stack-overflow report
  bool tuneBlocksAuto = true; // no RTAMD_BLOCKS_PER_CU given: 2 workgroups per CU, 1 when >= 2 batches run on other streams
  void memoryMonitor(ssize_t bytes, bool post);
};

// thread-local error slot used when no device exists yet (rtcNewDevice failure; device.cpp:262-263)
RTCError& thread_error();

// ---------------------------------------------------------------------------------------------------
// Buffer: owned or shared (borrowed) byte range.
// ---------------------------------------------------------------------------------------------------
struct Buffer : RefCounted
{
  Device* device;
  char* ptr = nullptr;
  size_t bytes = 0;
  bool shared = false;
  Buffer(Device* d, size_t n, void* sharedPtr);
  ~Buffer() override;
};

// A typed view into a Buffer as bound to a geometry slot (reference: RawBufferView, kernels/common/buffer.h).
struct BufferView
{
  Buffer* buf = nullptr;
  RTCFormat format = RTC_FORMAT_UNDEFINED;
  size_t offset = 0, stride = 0, count = 0;
  bool modified = true;
  const char* at(size_t i) const { return buf->ptr + offset + i * stride; }
  bool valid() const { return buf != nullptr; }
  void set(Buffer* b, RTCFormat f, size_t off, size_t str, size_t n);
  void clear();
};

// ---------------------------------------------------------------------------------------------------
// Geometry
// ---------------------------------------------------------------------------------------------------
struct Geometry : RefCounted
{
  Device* device;
  RTCGeometryType type;
  bool enabled = true;
  bool committed = false;
  unsigned mask = 0xFFFFFFFFu;
  unsigned timeSteps = 1;
  RTCBuildQuality quality = RTC_BUILD_QUALITY_MEDIUM;
  void* userPtr = nullptr;
  RTCFilterFunctionN intersectFilter = nullptr;
  RTCFilterFunctionN occludedFilter = nullptr;
  RTCDisplacementFunctionN displacement = nullptr;
  float tessellationRate = 2.0f;
  unsigned vertexAttribCount = 0;
  unsigned topologyCount = 1;
  std::vector<RTCSubdivisionMode> subdivMode{RTC_SUBDIVISION_MODE_SMOOTH_BOUNDARY};

  // buffer slots, keyed by (type, slot)
  std::map<std::pair<int, unsigned>, BufferView> views;

  // face-varying data: topology (= index buffer slot) used by each vertex attribute slot (default 0)
  std::vector<unsigned> attribTopology;
  // rtcInterpolate on subdivision meshes: refined buffers, built on first use, dropped by rtcCommitGeometry
  std::shared_ptr<void> interpCache;
  std::mutex interpMutex;

  Geometry(Device* d, RTCGeometryType t);
  ~Geometry() override;

  BufferView* view(RTCBufferType t, unsigned slot);
  const BufferView* view(RTCBufferType t, unsigned slot) const;
  void bind(RTCBufferType t, unsigned slot, RTCFormat f, Buffer* b, size_t off, size_t stride, size_t count);

  // triangle mesh accessors (reference: TriangleMesh, kernels/common/scene_triangle_mesh.h)
  size_t numTriangles() const;
  size_t numVertices() const;
  void triangle(size_t i, unsigned idx[3]) const;
  V3 vertex(size_t i) const;
  bool validTriangle(size_t i) const;
};

// ---------------------------------------------------------------------------------------------------
// Scene
// ---------------------------------------------------------------------------------------------------
// One device accel: host mirror (for tests / stats) + HBM arrays.
struct Accel
{
  std::vector<QNode8> nodes;
  std::vector<TriRecord> prims;
  std::vector<uint8_t> blobs;
  std::vector<uint32_t> blobOffsets;
  uint32_t root = REF_EMPTY;
  uint32_t kind = ACCEL_NONE;
  uint32_t robust = 0;
  uint32_t maxDepth = 0;
  uint32_t blobStride = 0;
  size_t leafCount = 0;
  // device copies, one set per shard of the device (replicated accel)
  struct DevCopy
  {
    void* dNodes = nullptr;
    void* dPrims = nullptr;
    void* dBlobs = nullptr;
    void* dBlobOffsets = nullptr;
  };
  std::vector<DevCopy> dev;
  std::vector<int> devOrdinals; // ordinal each copy lives on (for freeDevice)
  AccelDesc desc(size_t shard = 0) const;
  size_t deviceBytes() const;
  void upload(Device* dev);
  void freeDevice();
  void clear();
};

struct Scene : RefCounted
{
  Device* device;
  std::vector<Geometry*> geometries; // index = geomID, nullptr = free slot
  RTCSceneFlags flags = RTC_SCENE_FLAG_NONE;
  RTCBuildQuality quality = RTC_BUILD_QUALITY_MEDIUM;
  unsigned subdivisionLevel = 6; // fork defaults, kernels/common/scene.cpp:41-42
  unsigned compressionLevel = 3;
  RTCProgressMonitorFunction progressFn = nullptr;
  void* progressUser = nullptr;
  bool modified = true; // "scene got not committed" until the first commit (scene.cpp:25,54)
  // filter callbacks present at commit time (Scene::hasGeometryFilterFunction, scene.h): they route a batch through the
  // host filter loop of rt_trace.cpp
  bool triIntersectFilter = false, triOccludedFilter = false, subdivFilter = false;
  std::mutex buildMutex;
  Box3 bounds;

  std::vector<uint8_t> debugGrids; // keep_grids=1: per patch {geomID,primID,n} + x[],y[],z[] of the (n+1)^2 grid

  Accel triAccel;    // triangles
  Accel subdivAccel; // subdivision patches (cBVH / GridSOA leaves)

  explicit Scene(Device* d);
  ~Scene() override;

  unsigned attach(Geometry* g);
  void attachByID(Geometry* g, unsigned id);
  void detach(unsigned id);
  Geometry* get(unsigned id) const;
  void commit();
  bool isRobust() const { return (flags & RTC_SCENE_FLAG_ROBUST) != 0; }
};

} // namespace rtamd
