// Device: config parsing, error funnel, HIP device / stream / staging management.
#include <algorithm>
#include <cstdlib>
#include <pthread.h>

#include "rt_objects.h"
#include "rt_trace.h"

namespace rtamd {

RTCError& thread_error()
{
  static thread_local RTCError e = RTC_ERROR_NONE;
  return e;
}

static uint64_t this_thread_key() { return (uint64_t)pthread_self(); }

// Tokenizer for "key=value,key=value" strings; separators are ',' and whitespace, like the reference's
// TokenStream-driven State::parse (kernels/common/state.cpp:241-430).  Unknown keys are skipped.
void Device::parse(const std::string& cfg)
{
  size_t i = 0;
  auto isSep = [](char c) { return c == ',' || c == ' ' || c == '\t' || c == '\n' || c == ';'; };
  while (i < cfg.size()) {
    while (i < cfg.size() && isSep(cfg[i])) i++;
    size_t k0 = i;
    while (i < cfg.size() && !isSep(cfg[i]) && cfg[i] != '=') i++;
    std::string key = cfg.substr(k0, i - k0);
    std::string val;
    if (i < cfg.size() && cfg[i] == '=') {
      i++;
      size_t v0 = i;
      while (i < cfg.size() && !isSep(cfg[i])) i++;
      val = cfg.substr(v0, i - v0);
    }
    if (key.empty()) continue;
    if (key == "tri_accel" || key == "accel") tri_accel = val;
    else if (key == "subdiv_accel") subdiv_accel = val;
    else if (key == "verbose") verbose = atoi(val.c_str());
    else if (key == "gpu" || key == "device") { gpu = (val == "none") ? -1 : atoi(val.c_str()); gpuList.clear(); }
    else if (key == "gpus") {
      // "gpus=0-7" (range), "gpus=0:2:5" (list; ',' separates config keys), "gpus=0:0" (two logical shards on one GPU)
      // malformed input is an error, not a silent fall-back to gpu=0: empty list, inverted range, non-numeric token, > 64 shards
      gpuList.clear();
      auto number = [](const std::string& t) { return !t.empty() && t.find_first_not_of("0123456789") == std::string::npos; };
      size_t a = 0;
      while (a <= val.size()) {
        size_t b = val.find(':', a);
        if (b == std::string::npos) b = val.size();
        const std::string tok = val.substr(a, b - a);
        const size_t dash = tok.find('-');
        if (dash != std::string::npos) {
          const std::string sl = tok.substr(0, dash), sh = tok.substr(dash + 1);
          if (!number(sl) || !number(sh)) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "gpus=: malformed range '" + tok + "'");
          const int lo = atoi(sl.c_str()), hi = atoi(sh.c_str());
          if (hi < lo) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "gpus=: inverted range '" + tok + "'");
          for (int g = lo; g <= hi; g++) gpuList.push_back(g);
        } else {
          if (!number(tok)) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "gpus=: malformed entry '" + tok + "'");
          gpuList.push_back(atoi(tok.c_str()));
        }
        if (gpuList.size() > 64) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "gpus=: more than 64 shards");
        a = b + 1;
      }
      gpu = gpuList[0];
    }
    else if (key == "threads") numThreads = atoi(val.c_str());
    else if (key == "service") tuneService = (uint32_t)std::max(0, atoi(val.c_str())); // persistent consumer for calls of up to 64 rays (also env RTAMD_SERVICE)
    else if (key == "host_threads") tuneHostThreads = (uint32_t)std::max(0, atoi(val.c_str())); // staging threads of pipelined host batches (also env RTAMD_HOST_THREADS)
    else if (key == "benchmark") benchmark = atoi(val.c_str());
    else if (key == "keep_grids") keepGrids = atoi(val.c_str());
    // isa, max_isa, set_affinity, affinity, start_threads, hugepages, float_exceptions, ... : x86-only, ignored
  }
}

Device::Device(const char* cfg)
{
  if (const char* env = getenv("RTAMD_GPU")) gpu = atoi(env);
  if (const char* env = getenv("RTAMD_CHUNK_BUSY")) tuneChunkBusy = (uint32_t)std::max(1, atoi(env));
  if (const char* env = getenv("RTAMD_CHUNK")) { tuneChunk = (uint32_t)std::max(1, atoi(env)); tuneChunkFixed = true; }
  if (const char* env = getenv("RTAMD_SERVICE")) tuneService = (uint32_t)std::max(0, atoi(env));
  if (const char* env = getenv("RTAMD_LEAF_BATCH")) tuneLeafBatch = (uint32_t)std::max(1, atoi(env));
  if (const char* env = getenv("RTAMD_REFILL_BATCH")) tuneRefillBatch = (uint32_t)std::max(1, atoi(env));
  if (const char* env = getenv("RTAMD_OCT_LEAF")) tuneOctLeaf = (uint32_t)std::max(0, atoi(env));
  if (const char* env = getenv("RTAMD_CULL")) tuneCull = (uint32_t)std::max(0, atoi(env));
  if (const char* env = getenv("RTAMD_CBVH_FORM")) tuneCbvhForm = strcmp(env, "quad") == 0 ? 0u : (strcmp(env, "lane") == 0 ? 1u : 2u);
  if (const char* env = getenv("RTAMD_OCT_STEPS")) tuneOctSteps = (uint32_t)std::max(1, atoi(env));
  if (const char* env = getenv("RTAMD_BUSY_BLOCKS")) tuneBusyBlocksOct = (uint32_t)std::max(1, atoi(env));
  if (const char* env = getenv("RTAMD_ALONE_BLOCKS")) tuneAloneBlocksOct = (uint32_t)std::max(1, atoi(env));
  if (const char* env = getenv("RTAMD_ZEROCOPY_MAX")) tuneZeroCopyMax = (uint32_t)std::max(0, atoi(env));
  if (const char* env = getenv("RTAMD_HOST_THREADS")) tuneHostThreads = (uint32_t)std::max(0, atoi(env));
  if (const char* env = getenv("RTAMD_PIPE_MIN")) tunePipeMinRays = (uint32_t)std::max(1, atoi(env));
  if (const char* env = getenv("RTAMD_PIPE_CHUNK")) tunePipeChunk = (uint32_t)std::max(64, atoi(env));
  if (const char* env = getenv("RTAMD_WALK_BATCH")) tuneWalkBatch = (uint32_t)std::max(1, atoi(env));
  if (const char* env = getenv("RTAMD_OCT_MAX")) tuneOctMax = (uint32_t)std::max(0, atoi(env));
  if (const char* env = getenv("RTAMD_KERNEL")) tunePoolKernel = strcmp(env, "pool") == 0 ? 1u : (strcmp(env, "lane") == 0 ? 0u : 2u);
  if (const char* env = getenv("RTAMD_BLOCKS_PER_CU")) { tuneBlocksPerCU = (uint32_t)std::max(0, atoi(env)); tuneBlocksAuto = false; }
  if (cfg) parse(cfg);
  if (gpu == -1) {
    // "gpu=none": host-only object model (build + inspect accels); every trace call raises
    // RTC_ERROR_INVALID_OPERATION.  There is deliberately no CPU traversal in this library.
    return;
  }
  int n = 0;
  HIP_CHECK(hipGetDeviceCount(&n));
  if (n <= 0) RT_THROW(RTC_ERROR_UNKNOWN, "no HIP device available: the traversal path has no CPU fallback");
  if (gpuList.empty()) gpuList.push_back(gpu);
  for (int g : gpuList)
    if (g < 0 || g >= n) RT_THROW(RTC_ERROR_INVALID_ARGUMENT, "gpu ordinal out of range");
  gpu = gpuList[0];
  for (int g : gpuList) {
    std::unique_ptr<GpuShard> sh(new GpuShard);
    sh->ordinal = g;
    HIP_CHECK(hipSetDevice(g));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, g));
    sh->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_CHECK(hipStreamCreateWithFlags(&sh->stream, hipStreamNonBlocking));
    sh->ownsStream = true;
    HIP_CHECK(hipMalloc(&sh->countersDev, 2 * (size_t)WAVE_LOG_CAPACITY * sizeof(WaveRecord)));
    HIP_CHECK(hipHostMalloc((void**)&sh->overflowHost, 64, hipHostMallocMapped));
    *sh->overflowHost = 0u;
    HIP_CHECK(hipHostGetDevicePointer((void**)&sh->overflowDev, sh->overflowHost, 0));
    for (LaunchCtx& c : sh->launchCtx) {
      HIP_CHECK(hipMalloc(&c.queues, 64 * 32 * 4)); // TRACE_QUEUES heads, one 128-byte line each (TRACE_QUEUE_STRIDE)
      HIP_CHECK(hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
    }
    if (verbose >= 1)
      fprintf(stderr, "embree3-amd: shard %zu on device %d (%s, %d CUs), tri_accel=%s subdiv_accel=%s\n", shards.size(), g, prop.name, sh->numCUs,
              tri_accel.c_str(), subdiv_accel.c_str());
    shards.push_back(std::move(sh));
  }
  HIP_CHECK(hipSetDevice(gpu));
}

Device::~Device()
{
  service_destroy(this);
  for (auto& shp : shards) {
    GpuShard& sh = *shp;
    hipSetDevice(sh.ordinal);
    if (sh.stream && sh.ownsStream) {
      hipStreamSynchronize(sh.stream);
      hipStreamDestroy(sh.stream);
    }
    for (hipStream_t ps : sh.pipeStream)
      if (ps) { hipStreamSynchronize(ps); hipStreamDestroy(ps); }
    for (hipEvent_t e : sh.pipeEvents) hipEventDestroy(e);
    if (sh.stageHost) hipHostFree(sh.stageHost);
    if (sh.stageDev) hipFree(sh.stageDev);
    for (LaunchCtx& c : sh.launchCtx) {
      if (c.done) {
        if (c.used) hipEventSynchronize(c.done);
        hipEventDestroy(c.done);
      }
      if (c.spill) hipFree(c.spill);
      if (c.survivors) hipFree(c.survivors);
      if (c.queues) hipFree(c.queues);
    }
    if (sh.countersDev) hipFree(sh.countersDev);
    if (sh.overflowHost) hipHostFree(sh.overflowHost);
  }
}

void Device::useDevice() const
{
  if (gpu < 0 || shards.empty()) RT_THROW(RTC_ERROR_INVALID_OPERATION, "device was created with gpu=none: no HIP device, no traversal");
  HIP_CHECK(hipSetDevice(gpu));
}

void Device::synchronize()
{
  useDevice();
  for (auto& sh : shards) {
    sh->use();
    HIP_CHECK(hipStreamSynchronize(sh->stream));
  }
  HIP_CHECK(hipSetDevice(gpu));
  for (auto& sh : shards) sh->checkOverflow();
}

void Device::GpuShard::use() const { HIP_CHECK(hipSetDevice(ordinal)); }

void Device::GpuShard::checkOverflow()
{
  if (overflowHost && *(volatile uint32_t*)overflowHost != 0u) {
    *(volatile uint32_t*)overflowHost = 0u;
    RT_THROW(RTC_ERROR_UNKNOWN, "traversal stack overflow: a stack entry was dropped, results of the batch are incomplete");
  }
}

void Device::setError(RTCError code, const char* msg)
{
  if (verbose >= 1) {
    static const char* names[] = {"No error", "Unknown error", "Invalid argument", "Invalid operation",
                                  "Out of memory", "Unsupported CPU", "Cancelled"};
    fprintf(stderr, "Embree: %s", (unsigned)code < 7 ? names[code] : "Invalid error code");
    if (msg) fprintf(stderr, ", (%s)", msg);
    fprintf(stderr, "\n");
  }
  if (errorFn) errorFn(errorFnUser, code, msg);
  std::lock_guard<std::mutex> g(errMutex);
  RTCError& slot = threadErrors[this_thread_key()];
  if (slot == RTC_ERROR_NONE) slot = code; // first error wins until it is read
}

RTCError Device::takeError()
{
  // a stack overflow reported by a kernel of an asynchronous (device-resident) batch surfaces here at the latest
  for (auto& sh : shards) {
    try { sh->checkOverflow(); } catch (const rtc_error& e) { setError(e.code, e.msg.c_str()); }
  }
  std::lock_guard<std::mutex> g(errMutex);
  auto it = threadErrors.find(this_thread_key());
  if (it == threadErrors.end()) return RTC_ERROR_NONE;
  RTCError e = it->second;
  it->second = RTC_ERROR_NONE;
  return e;
}

void Device::memoryMonitor(ssize_t bytes, bool post)
{
  if (memFn && bytes != 0) {
    if (!memFn(memFnUser, bytes, post)) {
      if (bytes > 0) RT_THROW(RTC_ERROR_OUT_OF_MEMORY, "memory monitor forced termination"); // device.cpp:288-298
    }
  }
}

void Device::GpuShard::ensureStaging(size_t bytes)
{
  if (bytes <= stageBytes) return;
  size_t want = stageBytes ? stageBytes : (size_t)1 << 20;
  while (want < bytes) want *= 2;
  if (stageHost) hipHostFree(stageHost);
  if (stageDev) hipFree(stageDev);
  stageHost = stageDev = nullptr;
  stageBytes = 0;
  HIP_CHECK(hipHostMalloc(&stageHost, want, hipHostMallocDefault));
  HIP_CHECK(hipMalloc(&stageDev, want));
  stageBytes = want;
}

Device::LaunchCtx& Device::GpuShard::acquireLaunchCtx(size_t spillBytesNeeded, unsigned* busyOther, hipStream_t onStream)
{
  hipStream_t stream = onStream ? onStream : this->stream; // the stream the launch goes to
  unsigned others = 0;
  hipStream_t otherStreams[NUM_LAUNCH_CTX];
  // first context whose last kernel has finished (back-to-back batches on one stream then cycle through two or three
  // contexts, and only those get a spill area); all busy: take the next one in turn and wait for it on the stream
  LaunchCtx* pick = nullptr;
  for (LaunchCtx& k : launchCtx) {
    if (!k.used) { if (!pick) pick = &k; continue; }
    const hipError_t q = hipEventQuery(k.done);
    if (q == hipSuccess) { k.used = false; if (!pick) pick = &k; continue; }
    if (q != hipErrorNotReady) HIP_CHECK(q);
    if (k.stream != stream) { // count every OTHER stream once: queued launches on one stream run one after the other
      bool seen = false;
      for (unsigned i = 0; i < others; i++) seen |= otherStreams[i] == k.stream;
      if (!seen) otherStreams[others++] = k.stream;
    }
  }
  if (busyOther) *busyOther = others;
  (void)hipGetLastError(); // hipErrorNotReady is not an error
  LaunchCtx& c = pick ? *pick : launchCtx[nextCtx++ % NUM_LAUNCH_CTX];
  if (spillBytesNeeded > c.spillBytes) {
    // grow the overflow areas of ALL contexts now (first batch after a commit with a deeper tree): an allocation
    // synchronises the device, so it must not happen again when the next context is first used in the middle of a
    // pipelined sequence of batches
    HIP_CHECK(hipDeviceSynchronize());
    for (LaunchCtx& k : launchCtx) {
      if (k.spill) hipFree(k.spill);
      k.spill = nullptr;
      k.spillBytes = 0;
      HIP_CHECK(hipMalloc(&k.spill, spillBytesNeeded));
      k.spillBytes = spillBytesNeeded;
      k.used = false;
    }
  }
  if (c.used) HIP_CHECK(hipStreamWaitEvent(stream, c.done, 0));
  c.used = true;
  c.stream = stream;
  return c;
}

// ---- HostPool ----------------------------------------------------------------------------------------
// One step of a polling loop: the CPU's spin hint, and after a while a yield, so that a helper (or the caller) that lost its core to
// the application's own threads does not keep another one busy for nothing
static inline void spin_relax(unsigned& spins)
{
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield");
#endif
  if (++spins >= 4096u) { std::this_thread::yield(); spins = 0; }
}

bool Device::HostPool::take(uint64_t gen, size_t n, size_t& i)
{
  uint64_t t = ticket.load(std::memory_order_acquire);
  for (;;) {
    if ((t >> 32) != (gen & 0xffffffffull) || (t & 0xffffffffull) >= n) return false;
    if (ticket.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel, std::memory_order_acquire)) { i = (size_t)(t & 0xffffffffull); return true; }
  }
}

void Device::HostPool::start(unsigned n)
{
  while (threads.size() < n)
    threads.emplace_back([this] {
      uint64_t seen = 0;
      for (;;) {
        {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [this] { return stop.load() || hot.load(); });
          if (stop.load()) return;
        }
        unsigned spins = 0;
        while (hot.load(std::memory_order_acquire)) {
          const uint64_t g = ticket.load(std::memory_order_acquire) >> 32;
          if (g == seen) { spin_relax(spins); continue; }
          seen = g;
          spins = 0;
          const std::function<void(size_t)>* f = job.load(std::memory_order_relaxed); // written before the ticket of generation g was published
          const size_t n = nParts.load(std::memory_order_relaxed);
          size_t i;
          while (take(g, n, i)) {
            (*f)(i);
            done.fetch_add(1, std::memory_order_acq_rel);
          }
        }
      }
    });
}

void Device::HostPool::begin()
{
  {
    std::lock_guard<std::mutex> g(m);
    hot.store(true, std::memory_order_release);
  }
  cv.notify_all();
}

void Device::HostPool::end() { hot.store(false, std::memory_order_release); }

void Device::HostPool::run(size_t parts, const std::function<void(size_t)>& f)
{
  if (parts == 0) return;
  const uint64_t g = ((ticket.load(std::memory_order_relaxed) >> 32) + 1) & 0xffffffffull;
  job.store(&f, std::memory_order_relaxed);
  nParts.store(parts, std::memory_order_relaxed);
  done.store(0, std::memory_order_relaxed);
  ticket.store(g << 32, std::memory_order_release); // publishes job / nParts
  size_t i;
  while (take(g, parts, i)) {
    f(i);
    done.fetch_add(1, std::memory_order_acq_rel);
  }
  unsigned spins = 0;
  while (done.load(std::memory_order_acquire) < parts) spin_relax(spins);
}

Device::HostPool::~HostPool()
{
  {
    std::lock_guard<std::mutex> g(m);
    stop.store(true);
    hot.store(false);
  }
  cv.notify_all();
  for (std::thread& t : threads) t.join();
}

// ---- Buffer ------------------------------------------------------------------------------------------
Buffer::Buffer(Device* d, size_t n, void* sharedPtr) : device(d), bytes(n)
{
  device->retain();
  if (sharedPtr) {
    ptr = (char*)sharedPtr;
    shared = true;
  } else {
    device->memoryMonitor((ssize_t)n, false);
    // 16 bytes of slack so that float3 vertices can be read with 16-byte loads (verify.cpp:2143-2148 convention)
    ptr = (char*)aligned_alloc(16, ((n + 15) / 16) * 16 + 16);
    if (!ptr) RT_THROW(RTC_ERROR_OUT_OF_MEMORY, "buffer allocation failed");
    memset(ptr, 0, ((n + 15) / 16) * 16 + 16);
  }
}

Buffer::~Buffer()
{
  if (!shared && ptr) {
    free(ptr);
    try { device->memoryMonitor(-(ssize_t)bytes, true); } catch (...) {}
  }
  device->release();
}

void BufferView::set(Buffer* b, RTCFormat f, size_t off, size_t str, size_t n)
{
  if (b) b->retain();
  if (buf) buf->release();
  buf = b;
  format = f;
  offset = off;
  stride = str;
  count = n;
  modified = true;
}

void BufferView::clear() { set(nullptr, RTC_FORMAT_UNDEFINED, 0, 0, 0); }

} // namespace rtamd
