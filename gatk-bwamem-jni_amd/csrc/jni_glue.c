/*
 * jni_glue.c -- the nine JNI entry points + JNI_OnLoad of class
 * org.broadinstitute.hellbender.utils.bwa.BwaMemIndex, bound to the C ABI of include/bwamem_hip.h.
 *
 * Same symbols, signatures and ownership rules as the reference's glue
 * (src/main/c/org_broadinstitute_hellbender_utils_bwa_BwaMemIndex.c:43-165, init.c:12-29), so the unchanged
 * Java classes load this library through -DLIBBWA_PATH or as the jar resource /libbwa.Linux.so.
 * Needs <jni.h>: `make jni` where JAVA_HOME is set.  This build image has no JDK; the tests compile the file against
 * tests/jni_stub/jni.h (the slice of the JNI it uses, function tables at the specification's indices) and drive
 * JNI_OnLoad and all nine entry points through a fake JNIEnv (tests/jni_stub/jni_driver.c, tests/test_jni_glue.py):
 * response bytes against the C ABI's, the BwaMemPairEndStats -> slot 1 mapping, the NULL-buffer and free paths.
 */
#include <jni.h>
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "../../include/bwamem_hip.h"

/* mem_pestat_t as the reference fills it (...BwaMemIndex.c:21-40) */
typedef struct { int low, high; int failed; double avg, std; } pestat_t;

static jfieldID peStatClass_failedID, peStatClass_lowID, peStatClass_highID, peStatClass_averageID, peStatClass_stdID;

jint JNI_OnLoad(JavaVM* vm, void* reserved)
{
    JNIEnv* env;
    jclass cls;
    (void)reserved;
    if ((*vm)->GetEnv(vm, (void**)&env, JNI_VERSION_1_8) != JNI_OK) return JNI_ERR;
    if (!(cls = (*env)->FindClass(env, "org/broadinstitute/hellbender/utils/bwa/BwaMemPairEndStats"))) return JNI_ERR;
    if (!(peStatClass_failedID = (*env)->GetFieldID(env, cls, "failed", "Z"))) return JNI_ERR;
    if (!(peStatClass_lowID = (*env)->GetFieldID(env, cls, "low", "I"))) return JNI_ERR;
    if (!(peStatClass_highID = (*env)->GetFieldID(env, cls, "high", "I"))) return JNI_ERR;
    if (!(peStatClass_averageID = (*env)->GetFieldID(env, cls, "average", "D"))) return JNI_ERR;
    if (!(peStatClass_stdID = (*env)->GetFieldID(env, cls, "std", "D"))) return JNI_ERR;
    (*env)->DeleteLocalRef(env, cls);
    return JNI_VERSION_1_8;
}

static char* jstring_to_chars(JNIEnv* env, jstring in)
{
    const char* tmp = (*env)->GetStringUTFChars(env, in, 0);
    char* res = strdup(tmp);
    (*env)->ReleaseStringUTFChars(env, in, tmp);
    return res;
}

/* only the FR orientation (slot 1) can be supplied; the others are marked failed */
static int jobject_to_pestat(JNIEnv* env, jobject in, pestat_t* out)
{
    int i;
    if (in == NULL) return 0;
    memset(out, 0, sizeof(pestat_t) * 4);
    for (i = 0; i < 4; i++, out++) {
        if (i == 1) {
            out->failed = (int)(*env)->GetBooleanField(env, in, peStatClass_failedID);
            if (!out->failed) {
                out->low = (int)(*env)->GetIntField(env, in, peStatClass_lowID);
                out->high = (int)(*env)->GetIntField(env, in, peStatClass_highID);
                out->avg = (double)(*env)->GetDoubleField(env, in, peStatClass_averageID);
                out->std = (double)(*env)->GetDoubleField(env, in, peStatClass_stdID);
            }
        } else out->failed = 1;
    }
    return 1;
}

#define JNIFN(name) Java_org_broadinstitute_hellbender_utils_bwa_BwaMemIndex_##name

JNIEXPORT jboolean JNICALL JNIFN(createReferenceIndex)(JNIEnv* env, jclass cls, jstring jRef, jstring jPrefix, jstring jAlgo)
{
    char *ref = jstring_to_chars(env, jRef), *prefix = jstring_to_chars(env, jPrefix), *algo = jstring_to_chars(env, jAlgo);
    int rc = jnibwa_createReferenceIndex(ref, prefix, algo);
    (void)cls;
    if (rc == -1) {
        char message[256];
        jclass iae = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
        snprintf(message, sizeof message, "wrong algorithm name '%s'", algo);
        (*env)->ThrowNew(env, iae, message);
    }
    free(ref); free(prefix); free(algo);
    return rc == 0;
}

JNIEXPORT jboolean JNICALL JNIFN(createIndexImageFile)(JNIEnv* env, jclass cls, jstring jPrefix, jstring jImg)
{
    char *prefix = jstring_to_chars(env, jPrefix), *img = jstring_to_chars(env, jImg);
    jboolean res = !jnibwa_createIndexFile(prefix, img);
    (void)cls;
    free(prefix); free(img);
    return res;
}

JNIEXPORT jlong JNICALL JNIFN(openIndex)(JNIEnv* env, jclass cls, jstring jImg)
{
    char* fname = jstring_to_chars(env, jImg);
    int fd = open(fname, O_RDONLY);
    (void)cls;
    free(fname);
    if (fd == -1) return 0;
    return (jlong)(size_t)jnibwa_openIndex(fd);
}

JNIEXPORT jint JNICALL JNIFN(destroyIndex)(JNIEnv* env, jclass cls, jlong idxAddr)
{
    (void)env; (void)cls;
    if (!idxAddr) return 0;
    return jnibwa_destroyIndex((bwaidx_t*)(size_t)idxAddr);
}

JNIEXPORT jobject JNICALL JNIFN(createDefaultOptions)(JNIEnv* env, jclass cls)
{
    (void)cls;
    return (*env)->NewDirectByteBuffer(env, jnibwa_createDefaultOptions(), 168);
}

JNIEXPORT jobject JNICALL JNIFN(getRefContigNames)(JNIEnv* env, jclass cls, jlong idxAddr)
{
    size_t bufSize = 0;
    void* bufMem;
    jobject namesBuf;
    (void)cls;
    if (!idxAddr) return 0;
    bufMem = jnibwa_getRefContigNames((bwaidx_t*)(size_t)idxAddr, &bufSize);
    namesBuf = (*env)->NewDirectByteBuffer(env, bufMem, (jlong)bufSize);
    if (!namesBuf) jnibwa_free(bufMem);
    return namesBuf;
}

JNIEXPORT jobject JNICALL JNIFN(createAlignments)(JNIEnv* env, jclass cls, jobject seqsBuf, jlong idxAddr, jobject optsBuf, jobject frPEStats)
{
    pestat_t peStats[4];
    int provided = jobject_to_pestat(env, frPEStats, peStats);
    mem_opt_t* pOpts = (mem_opt_t*)(*env)->GetDirectBufferAddress(env, optsBuf);
    char* pSeq = (char*)(*env)->GetDirectBufferAddress(env, seqsBuf);
    size_t bufSize = 0;
    void* bufMem = jnibwa_createAlignments((bwaidx_t*)(size_t)idxAddr, pOpts, provided ? (mem_pestat_t*)peStats : 0, pSeq, &bufSize);
    jobject alnBuf;
    (void)cls;
    if (!bufMem) return 0;                       /* Java: IllegalStateException (BwaMemIndex.java:411-414) */
    alnBuf = (*env)->NewDirectByteBuffer(env, bufMem, (jlong)bufSize);
    if (!alnBuf) jnibwa_free(bufMem);
    return alnBuf;
}

JNIEXPORT void JNICALL JNIFN(destroyByteBuffer)(JNIEnv* env, jclass cls, jobject alnBuf)
{
    (void)cls;
    jnibwa_free((*env)->GetDirectBufferAddress(env, alnBuf));
}

JNIEXPORT jstring JNICALL JNIFN(getVersion)(JNIEnv* env, jclass cls)
{
    (void)cls;
    return (*env)->NewStringUTF(env, jnibwa_getVersion());
}
