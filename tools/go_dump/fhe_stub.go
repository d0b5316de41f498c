//go:build !withfhe

package main

import (
	"github.com/tuneinsight/lattigo/v6/core/rlwe"
	"github.com/tuneinsight/lattigo/v6/schemes/bgv"
)

// without -tags withfhe nothing of package fhe (and so nothing of lazer / cgo) is linked
func dumpFHE(string, bgv.Parameters, *rlwe.SecretKey, *rlwe.PublicKey, *bgv.Encoder, *rlwe.Encryptor) {}
