/*
 * tests/jni_stub/jni.h -- TEST INFRASTRUCTURE.  This image has no JDK, so csrc/jni_glue.c could never be compiled.
 * This header declares just the slice of the Java Native Interface that file (and the reference's glue,
 * src/main/c/org_broadinstitute_hellbender_utils_bwa_BwaMemIndex.c + init.c) uses, with the function tables laid out at
 * the indices of the JNI specification's "Interface Function Table" (written from the specification as remembered; no JDK
 * header is available offline to diff against), so that the object code compiled here indexes the table as a JVM's would.
 * The fake JNIEnv that implements these slots is tests/jni_stub/jni_driver.c.  Never part of the product.
 */
#ifndef JNI_STUB_H_
#define JNI_STUB_H_
#include <stdint.h>
#include <stdarg.h>

typedef uint8_t jboolean;
typedef int8_t jbyte;
typedef int32_t jint;
typedef int64_t jlong;
typedef double jdouble;
typedef jint jsize;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jthrowable;
struct _jfieldID;
typedef struct _jfieldID* jfieldID;

#define JNI_OK 0
#define JNI_ERR (-1)
#define JNI_FALSE 0
#define JNI_TRUE 1
#define JNI_VERSION_1_8 0x00010008
#define JNIEXPORT __attribute__((visibility("default")))
#define JNIIMPORT
#define JNICALL

struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNIInvokeInterface_;
typedef const struct JNIInvokeInterface_* JavaVM;

struct JNINativeInterface_ {
    void* reserved0_3[4];                                                       /*   0 ..   3 */
    void* slot4_5[2];                                                           /*   4 GetVersion, 5 DefineClass */
    jclass (JNICALL *FindClass)(JNIEnv*, const char*);                          /*   6 */
    void* slot7_13[7];                                                          /*   7 ..  13 */
    jint (JNICALL *ThrowNew)(JNIEnv*, jclass, const char*);                     /*  14 */
    void* slot15_22[8];                                                         /*  15 ..  22 */
    void (JNICALL *DeleteLocalRef)(JNIEnv*, jobject);                           /*  23 */
    void* slot24_93[70];                                                        /*  24 ..  93 */
    jfieldID (JNICALL *GetFieldID)(JNIEnv*, jclass, const char*, const char*);  /*  94 */
    void* slot95;                                                               /*  95 GetObjectField */
    jboolean (JNICALL *GetBooleanField)(JNIEnv*, jobject, jfieldID);            /*  96 */
    void* slot97_99[3];                                                         /*  97 ..  99 */
    jint (JNICALL *GetIntField)(JNIEnv*, jobject, jfieldID);                    /* 100 */
    void* slot101_102[2];                                                       /* 101 .. 102 */
    jdouble (JNICALL *GetDoubleField)(JNIEnv*, jobject, jfieldID);              /* 103 */
    void* slot104_166[63];                                                      /* 104 .. 166 */
    jstring (JNICALL *NewStringUTF)(JNIEnv*, const char*);                      /* 167 */
    void* slot168;                                                              /* 168 GetStringUTFLength */
    const char* (JNICALL *GetStringUTFChars)(JNIEnv*, jstring, jboolean*);      /* 169 */
    void (JNICALL *ReleaseStringUTFChars)(JNIEnv*, jstring, const char*);       /* 170 */
    void* slot171_228[58];                                                      /* 171 .. 228 */
    jobject (JNICALL *NewDirectByteBuffer)(JNIEnv*, void*, jlong);              /* 229 */
    void* (JNICALL *GetDirectBufferAddress)(JNIEnv*, jobject);                  /* 230 */
    jlong (JNICALL *GetDirectBufferCapacity)(JNIEnv*, jobject);                 /* 231 */
    void* slot232_234[3];
};

struct JNIInvokeInterface_ {
    void* reserved0_2[3];                                                       /* 0 .. 2 */
    void* slot3_5[3];                                                           /* 3 DestroyJavaVM, 4 AttachCurrentThread, 5 DetachCurrentThread */
    jint (JNICALL *GetEnv)(JavaVM*, void**, jint);                              /* 6 */
    void* slot7;                                                                /* 7 AttachCurrentThreadAsDaemon */
};

#endif
