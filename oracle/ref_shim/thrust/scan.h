#pragma once
#include "execution_policy.h"
namespace thrust {
template <typename In, typename Out>
inline Out inclusive_scan(device_t, In first, In last, Out out) {
    if (first == last) return out;
    auto acc = *first; *out = acc;
    for (++first, ++out; first != last; ++first, ++out) { acc = acc + *first; *out = acc; }
    return out;
}
template <typename In, typename Out>
inline Out exclusive_scan(device_t, In first, In last, Out out) {
    if (first == last) return out;
    auto acc = *first; *out = 0;   // init 0; in-place safe: read before write
    for (++first, ++out; first != last; ++first, ++out) { auto v = *first; *out = acc; acc = acc + v; }
    return out;
}
}
