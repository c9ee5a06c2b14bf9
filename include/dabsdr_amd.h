/*
 * dabsdr_amd.h — the 24 entry points of the reference's dabsdr C API, as
 * exported by libdabsdr_amd.so (SONAME libdabsdr.so.4).
 *
 * A host application keeps compiling against the reference's own header
 * (reference: lib/linux_x86_64/dabsdr.h, MIT, (c) Petr Kopecky); this file
 * re-declares the same binary interface for the build of the drop-in library
 * and for tests/, and states for each entry point which reference declaration
 * it replaces.  Type layouts are ABI-identical to dabsdr.h:37-394; only the
 * names of the structs are shared with it, the text is not.
 *
 * Behaviour behind the ABI: the reference library decodes sample-by-sample on
 * one CPU thread.  This library pulls whole transmission frames through the
 * same input callback, converts them to s16 IQ, pushes them into a one-stream
 * dabx context (include/dabx.h) and raises the callbacks from its worker thread.
 */
#ifndef DABSDR_AMD_H
#define DABSDR_AMD_H

#include <stdint.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DABSDR_API __attribute__((visibility("default")))

typedef struct dabsdr_s *dabsdrHandle_t;                       /* dabsdr.h:37 */

typedef struct { uint8_t major, minor, patch, flags; } dabsdrVersion_t;   /* dabsdr.h:39-45 */

typedef union {                                                /* dabsdr.h:47-60 */
    uint8_t raw;
    uint8_t mp2DRC;
    struct {
        uint8_t mpeg_surr_cfg : 3;
        uint8_t ps_flag : 1;
        uint8_t aac_channel_mode : 1;
        uint8_t sbr_flag : 1;
        uint8_t dac_rate : 1;
        uint8_t conceal : 1;
    } bits;
} dabsdrAudioFrameHeader_t;

typedef enum dabsdrDecoderId_e {                               /* dabsdr.h:62-67 */
    DABSDR_ID_DATA = -1,
    DABSDR_ID_AUDIO_PRIMARY = 0,
    DABSDR_ID_AUDIO_SECONDARY = 1,
} dabsdrDecoderId_t;

typedef struct {                                               /* dabsdr.h:71-78 */
    dabsdrDecoderId_t id;
    uint8_t ASCTy;
    dabsdrAudioFrameHeader_t header;
    uint16_t auLen;
    const uint8_t *pAuData;
} dabsdrAudioCBData_t;

typedef struct {                                               /* dabsdr.h:81-86 */
    dabsdrDecoderId_t id;
    uint16_t len;
    const uint8_t *pData;
} dabsdrDynamicLabelCBData_t;

typedef struct {                                               /* dabsdr.h:89-96 */
    dabsdrDecoderId_t id;
    uint16_t SCId;
    uint16_t userAppType;
    uint16_t dgLen;
    const uint8_t *pDgData;
} dabsdrDataGroupCBData_t;

typedef enum dabsdrNotificationId_e {                          /* dabsdr.h:99-118 */
    DABSDR_NID_SYNC_STATUS = 1,
    DABSDR_NID_TUNE,
    DABSDR_NID_ENSEMBLE_INFO,
    DABSDR_NID_SERVICE_LIST,
    DABSDR_NID_SERVICE_COMPONENT_LIST,
    DABSDR_NID_USER_APP_UPDATE,
    DABSDR_NID_USER_APP_LIST,
    DABSDR_NID_SERVICE_SELECTION,
    DABSDR_NID_SERVICE_STOP,
    DABSDR_NID_PERIODIC,
    DABSDR_NID_XPAD_APP_START_STOP,
    DABSDR_NID_RECONFIGURATION,
    DABSDR_NID_RESET,
    DABSDR_NID_ANNOUNCEMENT_SUPPORT,
    DABSDR_NID_ANNOUNCEMENT_SWITCHING,
    DABSDR_NID_PTY,
    DABSDR_NID_TII,
} dabsdrNotificationId_t;

typedef enum dabsdrNotificationStatus_e {                      /* dabsdr.h:120-127 */
    DABSDR_NSTAT_SUCCESS = 0,
    DABSDR_NSTAT_GENERIC_ERROR = 1,
    DABSDR_NSTAT_SERVICE_NOT_FOUND,
    DABSDR_NSTAT_SERVICE_NOT_READY,
    DABSDR_NSTAT_SERVICE_NOT_SUPPORTED,
} dabsdrNotificationStatus_t;

typedef struct {                                               /* dabsdr.h:129-135 */
    dabsdrNotificationId_t nid;
    dabsdrNotificationStatus_t status;
    uint16_t len;
    const void *pData;
} dabsdrNotificationCBData_t;

typedef enum {                                                 /* dabsdr.h:138-143 */
    DABSDR_SYNC_LEVEL_NO_SYNC = 0,
    DABSDR_SYNC_LEVEL_ON_NULL = 1,
    DABSDR_SYNC_LEVEL_FIC = 3,
} dabsdrSyncLevel_t;

typedef struct {                                               /* dabsdr.h:145-149 */
    dabsdrSyncLevel_t syncLevel;
    int16_t snr10;
} dabsdrNtfSyncStatus_t;

typedef struct {                                               /* dabsdr.h:151-166; sizeof == 32 (radiocontrol.cpp:2407) */
    dabsdrSyncLevel_t syncLevel;
    int16_t snr10;
    int32_t freqOffset;
    uint32_t dateHoursMinutes;
    uint16_t secMsec;
    uint16_t fibErrorCntr;
    uint8_t mscCrcOkCntr;
    uint8_t mscCrcErrorCntr;
    uint16_t audioServiceBytes;
    uint16_t padBytes;
    uint16_t rsUncorrectableCntr;
    uint16_t rsBitErrors;
    uint16_t rsBytes;
} dabsdrNtfPeriodic_t;

#define DAB_LABEL_MAX_LENGTH (16)
typedef struct {                                               /* dabsdr.h:169-174 */
    char str[DAB_LABEL_MAX_LENGTH + 1];
    uint16_t charField;
    uint8_t charset;
} dabsdrLabel_t;

typedef struct {                                               /* dabsdr.h:176-190 */
    uint32_t sid;
    dabsdrLabel_t label;
    struct { uint8_t s; uint8_t d; } pty;
    uint8_t CAId;
} dabsdrServiceListItem_t;

typedef struct {                                               /* dabsdr.h:193-243 */
    uint8_t SCIdS;
    uint8_t SubChId;
    int16_t SubChAddr;
    uint16_t SubChSize;
    uint8_t protectionLevel;
    union { uint8_t uepIdx; uint8_t fecScheme; };
    uint8_t ps;
    uint8_t lang;
    uint8_t CAflag;
    dabsdrLabel_t label;
    uint8_t numUserApps;
    uint8_t TMId;
    union {
        struct { uint8_t ASCTy; uint16_t bitRate; } streamAudio;
        struct { uint8_t DSCTy; uint16_t bitRate; } streamData;
        struct { uint8_t DSCTy; uint16_t SCId; uint8_t DGflag; int16_t packetAddress; } packetData;
    };
} dabsdrServiceCompListItem_t;

typedef struct {                                               /* dabsdr.h:245-258 */
    uint16_t type;
    dabsdrLabel_t label;
    uint8_t dataLen;
    uint8_t data[23];
} dabsdrUserAppListItem_t;

typedef struct {                                               /* dabsdr.h:260-264 */
    uint8_t numServices;
    int (*getServiceListItem)(dabsdrHandle_t handle, uint8_t idx, dabsdrServiceListItem_t *pServiceListItem);
} dabsdrNtfServiceList_t;

typedef struct {                                               /* dabsdr.h:266-271 */
    uint32_t SId;
    uint8_t numServiceComponents;
    int (*getServiceComponentListItem)(dabsdrHandle_t handle, uint8_t scIdx, dabsdrServiceCompListItem_t *pServiceCompListItem);
} dabsdrNtfServiceComponentList_t;

typedef struct {                                               /* dabsdr.h:273-285 */
    uint32_t SId;
    uint16_t ASu;
    uint8_t numClusterIds;
    uint8_t clusterIds[7];
} dabsdrNtfAnnouncementSupport_t;

typedef struct { uint32_t SId; uint8_t SCIdS; } dabsdrNtfUserAppUpdate_t;   /* dabsdr.h:287-291 */

typedef struct {                                               /* dabsdr.h:293-299 */
    uint32_t SId;
    uint8_t SCIdS;
    uint8_t numUserApps;
    int (*getUserAppListItem)(dabsdrHandle_t handle, uint8_t uaIdx, dabsdrUserAppListItem_t *pUserAppListItem);
} dabsdrNtfUserAppList_t;

typedef struct {                                               /* dabsdr.h:301-318 */
    uint32_t frequency;
    union {
        uint32_t ueid;
        struct { uint32_t eid : 16; uint32_t ecc : 8; };
    };
    int8_t LTO;
    uint8_t intTable;
    uint8_t alarm;
    dabsdrLabel_t label;
} dabsdrNtfEnsemble_t;

typedef struct {                                               /* dabsdr.h:320-325 */
    uint32_t SId;
    uint8_t SCIdS;
    dabsdrDecoderId_t id;
} dabsdrNtfServiceSelection_t;

typedef enum dabsdrNtfResetFlags_e {                           /* dabsdr.h:327-331 */
    DABSDR_RESET_INIT = 0,
    DABSDR_RESET_NEW_EID = 1
} dabsdrNtfResetFlags_t;

typedef dabsdrNtfServiceSelection_t dabsdrNtfServiceStop_t;   /* dabsdr.h:333 */

typedef struct { uint8_t appType; int8_t start; } dabsdrNtfXpadAppStartStop_t;   /* dabsdr.h:335-339 */

typedef struct { uint8_t clusterId; uint8_t subChId; uint16_t ASwFlags; } dabsdrAsw_t;   /* dabsdr.h:341-346 */
typedef struct { dabsdrAsw_t asw[8]; } dabsdrNtfAnnouncementSwitching_t;                /* dabsdr.h:348-351 */

typedef struct { uint32_t SId; uint8_t s; uint8_t d; } dabsdrNtfPTy_t;                  /* dabsdr.h:353-357 */

typedef enum dabsdrTiiMode_e {                                 /* dabsdr.h:359-363 */
    DABSDR_TII_MODE_CONSERVATIVE = 0,
    DABSDR_TII_MODE_DEFAULT = 1,
    DABSDR_TII_NUM_MODES
} dabsdrTiiMode_t;

typedef enum dabsdrSpectrum_e {                                /* dabsdr.h:365-370 */
    DABSDR_SPECT_SIGNAL = 0,
    DABSDR_SPECT_NULL = 1,
    DABSDR_SPECT_TII = 2,
    DABSDR_SPECT_NUM_TYPES
} dabsdrSpectrum_t;

typedef struct { uint8_t main; uint8_t sub; float level; } dabsdrTii_t;                 /* dabsdr.h:372-376 */

typedef struct {                                               /* dabsdr.h:379-384 */
    uint8_t numIds;
    dabsdrTii_t id[24];
    int (*getSpectrumTii)(dabsdrHandle_t handle, float buffer[192]);
} dabsdrNtfTii_t;

/* the host fills `buffer` with numSamples complex samples (2*numSamples floats,
 * I,Q interleaved) before returning: dabsdr.h:387, src/input/inputdevice.cpp:70-131 */
typedef void (*dabsdrInputFunc_t)(float[], uint16_t);

typedef void (*dabsdrAudioCBFunc_t)(dabsdrAudioCBData_t *p, void *ctx);                 /* dabsdr.h:390 */
typedef void (*dabsdrDynamicLabelCBFunc_t)(dabsdrDynamicLabelCBData_t *p, void *ctx);   /* dabsdr.h:391 */
typedef void (*dabsdrDataGroupCBFunc_t)(dabsdrDataGroupCBData_t *p, void *ctx);         /* dabsdr.h:392 */
typedef void (*dabsdrSpectrumCBFunc_t)(const float *p, dabsdrSpectrum_t type, void *ctx); /* dabsdr.h:393 */
typedef void (*dabsdrNotificationCBFunc_t)(dabsdrNotificationCBData_t *p, void *ctx);   /* dabsdr.h:394 */

/* lifecycle — dabsdr.h:397-402, used at src/radiocontrol.cpp:76-93 */
DABSDR_API void dabsdr(dabsdrHandle_t handle);                 /* starts the worker thread, returns */
DABSDR_API uint8_t dabsdrInit(dabsdrHandle_t *handle);         /* 0 (EXIT_SUCCESS) when a GPU context exists */
DABSDR_API void dabsdrGetVersion(dabsdrVersion_t *version);
DABSDR_API void dabsdrDeinit(dabsdrHandle_t *handle);          /* after dabsdrRequest_Exit; nulls *handle */

/* input side — dabsdr.h:405-406 */
DABSDR_API void dabsdrRegisterInputFcn(dabsdrHandle_t handle, dabsdrInputFunc_t fcn);
DABSDR_API void dabsdrRegisterDummyInputFcn(dabsdrHandle_t handle, dabsdrInputFunc_t fcn);

/* output callbacks — dabsdr.h:409-413 */
DABSDR_API void dabsdrRegisterAudioCb(dabsdrHandle_t handle, dabsdrAudioCBFunc_t fcn, void *ctx);
DABSDR_API void dabsdrRegisterDynamicLabelCb(dabsdrHandle_t handle, dabsdrDynamicLabelCBFunc_t fcn, void *ctx);
DABSDR_API void dabsdrRegisterDataGroupCb(dabsdrHandle_t handle, dabsdrDataGroupCBFunc_t fcn, void *ctx);
DABSDR_API void dabsdrRegisterSignalSpectrumCb(dabsdrHandle_t handle, dabsdrSpectrumCBFunc_t fcn, void *ctx);
DABSDR_API void dabsdrRegisterNotificationCb(dabsdrHandle_t handle, dabsdrNotificationCBFunc_t fcn, void *ctx);

/* asynchronous requests — dabsdr.h:417-429; callers src/radiocontrol.h:539-559 */
DABSDR_API void dabsdrRequest_Tune(dabsdrHandle_t handle, uint32_t frequency);
DABSDR_API void dabsdrRequest_GetEnsemble(dabsdrHandle_t handle);
DABSDR_API void dabsdrRequest_GetServiceList(dabsdrHandle_t handle);
DABSDR_API void dabsdrRequest_GetServiceComponents(dabsdrHandle_t handle, uint32_t SId);
DABSDR_API void dabsdrRequest_GetUserAppList(dabsdrHandle_t handle, uint32_t SId, uint8_t SCIdS);
DABSDR_API void dabsdrRequest_GetAnnouncementSupport(dabsdrHandle_t handle, uint32_t SId);
DABSDR_API void dabsdrRequest_ServiceSelection(dabsdrHandle_t handle, uint32_t SId, uint8_t SCIdS, dabsdrDecoderId_t id);
DABSDR_API void dabsdrRequest_ServiceStop(dabsdrHandle_t handle, uint32_t SId, uint8_t SCIdS, dabsdrDecoderId_t id);
DABSDR_API void dabsdrRequest_XPadAppStart(dabsdrHandle_t handle, uint8_t appType, int8_t startRequest, dabsdrDecoderId_t id);
DABSDR_API void dabsdrRequest_SetPeriodicNotify(dabsdrHandle_t handle, uint8_t period, uint32_t cfg);
DABSDR_API void dabsdrRequest_SetTII(dabsdrHandle_t handle, uint8_t ena, dabsdrTiiMode_t mode);
DABSDR_API void dabsdrRequest_SignalSpectrum(dabsdrHandle_t handle, uint8_t ena);
DABSDR_API void dabsdrRequest_Exit(dabsdrHandle_t handle);

#ifdef __cplusplus
}
#endif
#endif
