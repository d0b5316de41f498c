/* TEST DOUBLE: a librccl.so.1 that lacks most of the entry points lm_group.hip resolves -- rccl_load() must report
 * "librccl lacks: ..." and LUMEN_TRANSPORT_AUTO must fall back to device copies (tests/test_group_rccl.py). */
int ncclGetVersion(int *version) {
    if (version) *version = 1;
    return 0;
}
