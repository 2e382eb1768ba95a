/* TEST-ONLY, see ../Rinternals.h */
void GetRNGstate(void);
void PutRNGstate(void);
double unif_rand(void);
