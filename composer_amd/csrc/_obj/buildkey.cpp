extern "C" const char* cmp_build_key(void) { return "89cfaa4aa28e89a3e257a06429174215543ed4aa6ee7c9802a0b9735a22e21d5"; }
