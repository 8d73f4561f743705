// cast_string.hip — numeric -> Utf8 / LargeUtf8 (placeholder until the device
// Ryu lands).
#include "common.hpp"

ah_status ah_cast_to_string(ah_context* ctx, const ah_array_view* values, ah_type to_type,
                            ah_array_out* out) {
  return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "cast %s -> %s", ah_type_name(values->type),
                 ah_type_name(to_type));
}
