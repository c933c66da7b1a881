// TEST INFRASTRUCTURE: sequential stand-ins for the in-thread thrust calls of the reference.
#pragma once
namespace thrust { struct device_t {}; static const device_t device = device_t(); }
