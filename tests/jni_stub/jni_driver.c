/*
 * tests/jni_stub/jni_driver.c -- TEST INFRASTRUCTURE: a fake JNIEnv / JavaVM (the slots of tests/jni_stub/jni.h) and a
 * driver that calls JNI_OnLoad and the nine Java_org_broadinstitute_hellbender_utils_bwa_BwaMemIndex_* entry points of
 * csrc/jni_glue.c the way BwaMemIndex.java does, against whichever library with the jnibwa_* ABI it is linked to (the
 * emulation build in the CPU suite, libbwamem_hip.so on the GPU box).  Checks (reference lines in parentheses):
 *   - createAlignments' bytes == jnibwa_createAlignments' bytes, single-end and paired-end          (...BwaMemIndex.c:115-141)
 *   - BwaMemPairEndStats -> orientation slot 1, the other three failed                               (...BwaMemIndex.c:21-40)
 *   - a NULL DirectByteBuffer frees the native block with the one allocator                          (...BwaMemIndex.c:104-107,152-155)
 *   - destroyByteBuffer frees with that allocator                                                    (...BwaMemIndex.c:157-160)
 *   - a wrong algorithm name throws IllegalArgumentException                                         (...BwaMemIndex.c:16-19,50-56)
 * usage: jni_driver <ref.fa> <scratch dir>
 */
#define _GNU_SOURCE
#include <jni.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/bwamem_hip.h"

#define JNIFN(name) Java_org_broadinstitute_hellbender_utils_bwa_BwaMemIndex_##name
jint JNI_OnLoad(JavaVM* vm, void* reserved);
jboolean JNIFN(createReferenceIndex)(JNIEnv*, jclass, jstring, jstring, jstring);
jboolean JNIFN(createIndexImageFile)(JNIEnv*, jclass, jstring, jstring);
jlong JNIFN(openIndex)(JNIEnv*, jclass, jstring);
jint JNIFN(destroyIndex)(JNIEnv*, jclass, jlong);
jobject JNIFN(createDefaultOptions)(JNIEnv*, jclass);
jobject JNIFN(getRefContigNames)(JNIEnv*, jclass, jlong);
jobject JNIFN(createAlignments)(JNIEnv*, jclass, jobject, jlong, jobject, jobject);
void JNIFN(destroyByteBuffer)(JNIEnv*, jclass, jobject);
jstring JNIFN(getVersion)(JNIEnv*, jclass);

/* ---- the fake objects */
enum { K_CLASS = 1, K_STRING, K_BUFFER, K_PESTAT };
struct _jobject { int kind; const char* name; char* chars; void* addr; jlong cap; int failed, low, high; double average, std; };
struct _jfieldID { const char* name; const char* sig; };
static struct _jfieldID f_failed = { "failed", "Z" }, f_low = { "low", "I" }, f_high = { "high", "I" }, f_average = { "average", "D" }, f_std = { "std", "D" };
static int n_live_utf = 0, n_throw = 0, fail_next_buffer = 0, n_free = 0, n_local_deleted = 0;
static char thrown_class[128], thrown_msg[256];
static void* last_freed = 0;

static jobject new_obj(int kind) { jobject o = (jobject)calloc(1, sizeof(struct _jobject)); o->kind = kind; return o; }
static jclass JNICALL fake_FindClass(JNIEnv* e, const char* name) { (void)e; jobject o = new_obj(K_CLASS); o->name = name; return o; }
static jint JNICALL fake_ThrowNew(JNIEnv* e, jclass c, const char* msg) { (void)e; ++n_throw; snprintf(thrown_class, sizeof thrown_class, "%s", c->name); snprintf(thrown_msg, sizeof thrown_msg, "%s", msg); return 0; }
static void JNICALL fake_DeleteLocalRef(JNIEnv* e, jobject o) { (void)e; ++n_local_deleted; free(o); }
static jfieldID JNICALL fake_GetFieldID(JNIEnv* e, jclass c, const char* name, const char* sig)
{
    (void)e;
    if (c->kind != K_CLASS || strcmp(c->name, "org/broadinstitute/hellbender/utils/bwa/BwaMemPairEndStats")) return 0;
    struct _jfieldID* all[5] = { &f_failed, &f_low, &f_high, &f_average, &f_std };
    for (int i = 0; i < 5; ++i) if (!strcmp(all[i]->name, name) && !strcmp(all[i]->sig, sig)) return all[i];
    return 0;                                                   /* (BwaMemPairEndStats.java: failed Z, low I, high I, average D, std D) */
}
static jboolean JNICALL fake_GetBooleanField(JNIEnv* e, jobject o, jfieldID f) { (void)e; if (o->kind != K_PESTAT || f != &f_failed) abort(); return (jboolean)o->failed; }
static jint JNICALL fake_GetIntField(JNIEnv* e, jobject o, jfieldID f) { (void)e; if (o->kind != K_PESTAT) abort(); if (f == &f_low) return o->low; if (f == &f_high) return o->high; abort(); }
static jdouble JNICALL fake_GetDoubleField(JNIEnv* e, jobject o, jfieldID f) { (void)e; if (o->kind != K_PESTAT) abort(); if (f == &f_average) return o->average; if (f == &f_std) return o->std; abort(); }
static jstring JNICALL fake_NewStringUTF(JNIEnv* e, const char* s) { (void)e; jobject o = new_obj(K_STRING); o->chars = strdup(s); return o; }
static const char* JNICALL fake_GetStringUTFChars(JNIEnv* e, jstring s, jboolean* copy) { (void)e; if (s->kind != K_STRING) abort(); if (copy) *copy = 1; ++n_live_utf; return strdup(s->chars); }
static void JNICALL fake_ReleaseStringUTFChars(JNIEnv* e, jstring s, const char* c) { (void)e; (void)s; --n_live_utf; free((void*)c); }
static jobject JNICALL fake_NewDirectByteBuffer(JNIEnv* e, void* addr, jlong cap)
{
    (void)e;
    if (fail_next_buffer) { fail_next_buffer = 0; return 0; }   /* what a JVM out of memory returns */
    jobject o = new_obj(K_BUFFER); o->addr = addr; o->cap = cap; return o;
}
static void* JNICALL fake_GetDirectBufferAddress(JNIEnv* e, jobject o) { (void)e; if (o->kind != K_BUFFER) abort(); return o->addr; }
static jlong JNICALL fake_GetDirectBufferCapacity(JNIEnv* e, jobject o) { (void)e; return o->cap; }

static struct JNINativeInterface_ env_table;
static JNIEnv the_env = &env_table;
static jint JNICALL fake_GetEnv(JavaVM* vm, void** penv, jint version) { (void)vm; if (version != JNI_VERSION_1_8) return JNI_ERR; *penv = (void*)&the_env; return JNI_OK; }
static struct JNIInvokeInterface_ vm_table;
static JavaVM the_vm = &vm_table;

/* the executable's own jnibwa_free comes first in symbol resolution: the glue's frees are counted, then forwarded */
void jnibwa_free(void* p)
{
    static void (*real)(void*);
    if (!real) real = (void (*)(void*))dlsym(RTLD_NEXT, "jnibwa_free");
    ++n_free; last_freed = p;
    real(p);
}

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "jni_driver: check failed at line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

static jobject direct(void* p, size_t n) { jobject o = new_obj(K_BUFFER); o->addr = p; o->cap = (jlong)n; return o; }

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: jni_driver <ref.fa> <scratch dir>\n"); return 2; }
    JNIEnv* env = &the_env;
    env_table.FindClass = fake_FindClass; env_table.ThrowNew = fake_ThrowNew; env_table.DeleteLocalRef = fake_DeleteLocalRef;
    env_table.GetFieldID = fake_GetFieldID; env_table.GetBooleanField = fake_GetBooleanField; env_table.GetIntField = fake_GetIntField;
    env_table.GetDoubleField = fake_GetDoubleField; env_table.NewStringUTF = fake_NewStringUTF; env_table.GetStringUTFChars = fake_GetStringUTFChars;
    env_table.ReleaseStringUTFChars = fake_ReleaseStringUTFChars; env_table.NewDirectByteBuffer = fake_NewDirectByteBuffer;
    env_table.GetDirectBufferAddress = fake_GetDirectBufferAddress; env_table.GetDirectBufferCapacity = fake_GetDirectBufferCapacity;
    vm_table.GetEnv = fake_GetEnv;
    /* the table offsets the compiled glue uses are those of the specification */
    CHECK(((char*)&env_table.FindClass - (char*)&env_table) / sizeof(void*) == 6 && ((char*)&env_table.GetFieldID - (char*)&env_table) / sizeof(void*) == 94);
    CHECK(((char*)&env_table.NewStringUTF - (char*)&env_table) / sizeof(void*) == 167 && ((char*)&env_table.NewDirectByteBuffer - (char*)&env_table) / sizeof(void*) == 229);

    CHECK(JNI_OnLoad(&the_vm, 0) == JNI_VERSION_1_8 && n_local_deleted == 1);

    char prefix[1024], img[1024];
    snprintf(prefix, sizeof prefix, "%s/jni_ref", argv[2]); snprintf(img, sizeof img, "%s/jni_ref.img", argv[2]);
    jstring jfa = fake_NewStringUTF(env, argv[1]), jprefix = fake_NewStringUTF(env, prefix), jimg = fake_NewStringUTF(env, img);
    /* wrong algorithm name: IllegalArgumentException, false */
    CHECK(JNIFN(createReferenceIndex)(env, 0, jfa, jprefix, fake_NewStringUTF(env, "bwtsw2")) == 0 && n_throw == 1);
    CHECK(!strcmp(thrown_class, "java/lang/IllegalArgumentException") && strstr(thrown_msg, "wrong algorithm name 'bwtsw2'"));
    CHECK(JNIFN(createReferenceIndex)(env, 0, jfa, jprefix, fake_NewStringUTF(env, "auto")) == 1 && n_throw == 1);
    CHECK(JNIFN(createIndexImageFile)(env, 0, jprefix, jimg) == 1);
    CHECK(JNIFN(createIndexImageFile)(env, 0, fake_NewStringUTF(env, "/nonexistent/prefix"), jimg) == 0);
    CHECK(n_live_utf == 0);
    CHECK(JNIFN(openIndex)(env, 0, fake_NewStringUTF(env, "/nonexistent/image")) == 0);
    const jlong idx = JNIFN(openIndex)(env, 0, jimg);
    CHECK(idx != 0);
    CHECK(JNIFN(destroyIndex)(env, 0, 0) == 0 && JNIFN(getRefContigNames)(env, 0, 0) == 0);

    jstring ver = JNIFN(getVersion)(env, 0);
    CHECK(ver && ver->kind == K_STRING && !strcmp(ver->chars, jnibwa_getVersion()));

    /* contig names: same bytes as the C ABI; a NULL buffer frees the block */
    size_t want_n = 0; void* want = jnibwa_getRefContigNames((bwaidx_t*)(size_t)idx, &want_n);
    jobject names = JNIFN(getRefContigNames)(env, 0, idx);
    CHECK(names && (size_t)names->cap == want_n && !memcmp(names->addr, want, want_n));
    int f0 = n_free;
    JNIFN(destroyByteBuffer)(env, 0, names);
    CHECK(n_free == f0 + 1 && last_freed == names->addr);
    fail_next_buffer = 1; f0 = n_free;
    CHECK(JNIFN(getRefContigNames)(env, 0, idx) == 0 && n_free == f0 + 1);

    jobject opts = JNIFN(createDefaultOptions)(env, 0);
    CHECK(opts && opts->cap == 168);
    mem_opt_t* ref_opts = jnibwa_createDefaultOptions();
    CHECK(!memcmp(opts->addr, ref_opts, 168));

    /* single-end call (request layout of BwaMemAligner.java:198-209) */
    static const char* reads[4] = {
        "GGCTTTTAATGCTTTTCAGTGGTTGCTGCTCAAGATGGAGTCTACTCAGCAGATGGTAAGCTCTATTATT",
        "TTGTTTTTAACACCAGAGTCATCCATCACATAATCAAATTTACTTTTAACTCTGGTAAATACTTCATTGT",
        "AATACTTCTTTTGAAGCTGCAGTTGTTGCTGCCTTCAACATTAGAATTAATGGGTATTCAATATGATT", "ACGTACGTACGTACGTACGTACGTACGT" };
    char req[2048]; size_t rl = 4; *(int32_t*)req = 4;
    for (int i = 0; i < 4; ++i) { size_t l = strlen(reads[i]) + 1; memcpy(req + rl, reads[i], l); rl += l; }
    size_t n1 = 0; void* r1 = jnibwa_createAlignments((bwaidx_t*)(size_t)idx, (mem_opt_t*)opts->addr, 0, req, &n1);
    CHECK(r1 && n1 > 16);
    jobject a1 = JNIFN(createAlignments)(env, 0, direct(req, rl), idx, opts, 0);
    CHECK(a1 && (size_t)a1->cap == n1 && !memcmp(a1->addr, r1, n1));
    JNIFN(destroyByteBuffer)(env, 0, a1);
    fail_next_buffer = 1; f0 = n_free;
    CHECK(JNIFN(createAlignments)(env, 0, direct(req, rl), idx, opts, 0) == 0 && n_free == f0 + 1);

    /* paired-end with BwaMemPairEndStats: slot 1 = the object's fields, slots 0, 2, 3 failed */
    *(int32_t*)((char*)opts->addr + 60) |= 0x2;                 /* MEM_F_PE (BwaMemAligner.java:73) */
    *(int32_t*)req = 2;
    jobject pes = new_obj(K_PESTAT); pes->failed = 0; pes->low = 1; pes->high = 600; pes->average = 200.0; pes->std = 10.0;
    struct { int low, high, failed, pad; double avg, std; } want_pes[4];
    memset(want_pes, 0, sizeof want_pes);
    for (int i = 0; i < 4; ++i) want_pes[i].failed = 1;
    want_pes[1].failed = 0; want_pes[1].low = 1; want_pes[1].high = 600; want_pes[1].avg = 200.0; want_pes[1].std = 10.0;
    size_t n2 = 0; void* r2 = jnibwa_createAlignments((bwaidx_t*)(size_t)idx, (mem_opt_t*)opts->addr, (mem_pestat_t*)want_pes, req, &n2);
    jobject a2 = JNIFN(createAlignments)(env, 0, direct(req, rl), idx, opts, pes);
    CHECK(r2 && a2 && (size_t)a2->cap == n2 && !memcmp(a2->addr, r2, n2));
    /* a failed statistics object: every orientation failed */
    pes->failed = 1;
    for (int i = 0; i < 4; ++i) { memset(&want_pes[i], 0, sizeof want_pes[i]); want_pes[i].failed = 1; }
    size_t n3 = 0; void* r3 = jnibwa_createAlignments((bwaidx_t*)(size_t)idx, (mem_opt_t*)opts->addr, (mem_pestat_t*)want_pes, req, &n3);
    jobject a3 = JNIFN(createAlignments)(env, 0, direct(req, rl), idx, opts, pes);
    CHECK(r3 && a3 && (size_t)a3->cap == n3 && !memcmp(a3->addr, r3, n3));
    /* no statistics object: inferred per call (NULL at jnibwa.c:214) */
    size_t n4 = 0; void* r4 = jnibwa_createAlignments((bwaidx_t*)(size_t)idx, (mem_opt_t*)opts->addr, 0, req, &n4);
    jobject a4 = JNIFN(createAlignments)(env, 0, direct(req, rl), idx, opts, 0);
    CHECK(r4 && a4 && (size_t)a4->cap == n4 && !memcmp(a4->addr, r4, n4));

    JNIFN(destroyByteBuffer)(env, 0, a2); JNIFN(destroyByteBuffer)(env, 0, a3); JNIFN(destroyByteBuffer)(env, 0, a4); JNIFN(destroyByteBuffer)(env, 0, opts);
    jnibwa_free(r1); jnibwa_free(r2); jnibwa_free(r3); jnibwa_free(r4); jnibwa_free(want); jnibwa_free(ref_opts);
    CHECK(JNIFN(destroyIndex)(env, 0, idx) == 0);
    CHECK(n_live_utf == 0);
    printf("jni-glue-ok: 9 entry points + JNI_OnLoad, %zu + %zu + %zu + %zu response bytes identical to the C ABI\n", n1, n2, n3, n4);
    return 0;
}
