extern "C" const char* cmp_build_key(void) { return "4155e28bae357178bab751d9838f32fc3a0a6f75818d079f811e4819cd56636f"; }
