// vpn_api.hip — ABI version and error strings of libvpn_hip.so
#include "vpn_common.h"

extern "C" int vpn_abi_version(void) { return VPN_ABI_VERSION; }

extern "C" const char* vpn_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case VPN_E_BADARG: return "vpn: null pointer or non-positive size";
        case VPN_E_TOOBIG: return "vpn: size above a documented limit";
        case VPN_E_KIND: return "vpn: unknown primitive kind";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "vpn: unknown error code";
}
