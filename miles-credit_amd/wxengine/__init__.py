"""wxengine: host side of the MI355X-native CrossFormer / WXFormer forecast-step engine.

Everything here sits above the C ABI of include/wxengine.h (libwxengine.so, built by __graft_entry__.build()):
  engine    ctypes binding: WXEngine, WXPostBlock, DevicePreblock handles
  model     the registry-facing nn.Module (reference constructor kwargs and state-dict key names)
  rollout   the autoregressive step loop;  latband: one forecast sharded over ranks by latitude
  config / synth    model geometry, name-keyed synthetic weights and inputs for tests and the benchmark
There is no CPU fallback: without the HIP library or a GPU, construction raises.
"""
