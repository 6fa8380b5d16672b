/* Empty stand-in for <curand_kernel.h> (utils.h:7 includes it; nothing on the host path uses it). */
#pragma once
