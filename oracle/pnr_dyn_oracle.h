/* pnr_dyn_oracle.h — float64 restatement of the dynamics-mode step (ABA + PD).
 * TEST INFRASTRUCTURE ONLY (see pnr_oracle.h).  Filled in with the dynamics kernel. */
#ifndef PNR_DYN_ORACLE_H
#define PNR_DYN_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif
int orc_dyn_available(void);
#ifdef __cplusplus
}
#endif
#endif
