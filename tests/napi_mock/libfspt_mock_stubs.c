/* libfspt_mock_stubs.c - the rest of the C ABI the addon links against, for tests/napi_mock/libfspt_mock.c (no prototypes
 * here: these are never called by the handle tests) */
#define STUB(name) int name() { return -100; }
STUB(fspt_builder_autofocus) STUB(fspt_builder_build) STUB(fspt_builder_commit_obj) STUB(fspt_builder_counts) STUB(fspt_builder_get)
STUB(fspt_builder_group_info) STUB(fspt_builder_mtllib_name) STUB(fspt_builder_parse_obj) STUB(fspt_enable_counters) STUB(fspt_env_bins)
STUB(fspt_get_counters) STUB(fspt_set_texture_interleave_budget) STUB(fspt_target_path_state_bytes) STUB(fspt_target_prepare)
STUB(fspt_target_set_deferred) STUB(fspt_target_set_memory_limit) STUB(fspt_target_set_pipeline) STUB(fspt_target_set_pool)
STUB(fspt_target_set_shard) STUB(fspt_target_set_stage_timing) STUB(fspt_target_set_tail) STUB(fspt_target_set_trace_budget)
STUB(fspt_target_set_viewport) STUB(fspt_device_memory)
