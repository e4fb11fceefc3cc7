// OhTypes.h -- the handful of ohNet vocabulary types the pipeline's Msg model is written in
// (TUint, TByte, Brx/Brn, ASSERT -> AssertionFailed, EXCEPTION), provided here because ohNet is an
// external dependency of the reference and is not part of this repository.  Same names and the same
// error behaviour as the reference relies on: ASSERT throws AssertionFailed (the reference's tests use
// TEST_THROWS(..., AssertionFailed)), invalid sample rates throw SampleRateInvalid (Msg.cpp:472).
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <exception>
#include <string>

namespace OpenHome {

typedef uint32_t TUint;
typedef int32_t TInt;
typedef uint8_t TByte;
typedef uint16_t TUint16;
typedef int16_t TInt16;
typedef uint64_t TUint64;
typedef int64_t TInt64;
typedef bool TBool;
typedef char TChar;

class Exception : public std::exception {
public:
    Exception(const char* aName, const char* aFile, int aLine) : iMsg(std::string(aName) + " at " + aFile + ":" + std::to_string(aLine)) {}
    const char* what() const noexcept override { return iMsg.c_str(); }
private:
    std::string iMsg;
};

#define OH_EXCEPTION(Name)                                                                 \
    class Name : public ::OpenHome::Exception {                                           \
    public:                                                                                \
        Name(const char* aFile, int aLine) : ::OpenHome::Exception(#Name, aFile, aLine) {} \
    }
#define THROW(Name) throw Name(__FILE__, __LINE__)

OH_EXCEPTION(AssertionFailed);

#define ASSERT(x)  do { if (!(x)) { THROW(::OpenHome::AssertionFailed); } } while (0)
#define ASSERTS()  THROW(::OpenHome::AssertionFailed)

// Read-only view of bytes (ohNet's Brx/Brn); borrowed for the duration of a call.
class Brx {
public:
    Brx() : iPtr(nullptr), iBytes(0) {}
    Brx(const TByte* aPtr, TUint aBytes) : iPtr(aPtr), iBytes(aBytes) {}
    const TByte* Ptr() const { return iPtr; }
    TUint Bytes() const { return iBytes; }
    TByte At(TUint aIndex) const { ASSERT(aIndex < iBytes); return iPtr[aIndex]; }
    TByte operator[](TUint aIndex) const { return At(aIndex); }
    void Set(const TByte* aPtr, TUint aBytes) { iPtr = aPtr; iBytes = aBytes; }
protected:
    const TByte* iPtr;
    TUint iBytes;
};
typedef Brx Brn;

}  // namespace OpenHome
